// Fused feature kernel: int16 PCM -> 5 feature rows at 210 frames/s (gfx950).
//
// Replaces get_energy / get_zero_crossings / downsample_blur / get_freq_bands
// (describealign.py:545-593).  One pass over the PCM: a workgroup stages a chunk of frames
// (plus an 8-frame halo either side) from HBM into LDS with 16-byte loads, rounding through
// float16 exactly as the reference's array does (:156), and everything downstream runs out of
// LDS:
//   stage 1  one thread per 70 samples (two 35-sample groups; the window always starts on an odd
//            half-word, so every lane unpacks the same way): block-energy partials with packed
//            half dot products, sign changes on packed words, the 5x3 low-pass (bb1), band-1
//            residual energy (be1) and its three separable blur sums
//   stage 2  one thread per 35-sample group: the 7x3 low-pass (bb2), band-2 residual energy,
//            band-3 energy
//   stage 3  one thread per frame: gather the six groups of the frame
//   stage 4  one thread per output frame: 13-tap Hann smoothing of energy / zero crossings,
//            15-tap combination of the blur sums, log10(1+x)/2, store.
// The stage outputs are written over the PCM region once every thread is done reading it, so a
// workgroup needs 27 KB of LDS and four fit per CU (register-limited).
// The 630-tap and 90-tap Hann blurs of the reference are evaluated in their exactly equivalent
// separable form (see FeatTables): per-group partial sums in float32, combined in float64.
//
// Roofline: HBM bound.  Algorithmic bytes = 2*C*N read + 5 rows * 4 B * N/210 written.
#include "dalign_common.h"

namespace da {

typedef _Float16 half2_t __attribute__((ext_vector_type(2)));

#ifndef DA_FEAT_MIN_WAVES
#define DA_FEAT_MIN_WAVES 1
#endif
#ifndef DA_FEAT_EXT_MONO
#define DA_FEAT_EXT_MONO 64      /* 64-frame chunks: 27 KB of LDS, four workgroups per CU; 128-frame chunks (two per CU) were 25 % slower */
#endif
template <int C> struct FeatCfg {
  static constexpr int kExt = (C == 1) ? DA_FEAT_EXT_MONO : 64;    // frames staged per workgroup (incl. halo)
  static constexpr int kHalo = 8;
  static constexpr int kOut = kExt - 2 * kHalo;       // frames produced per workgroup
  static constexpr int kNQ = kExt * 6;                // 35-sample groups per chunk
  static constexpr int kPairs = kNQ / 2;              // stage-1 work items = threads per workgroup
  static constexpr int kThreads = kPairs;             // 384 (mono) / 192 (stereo)
  static constexpr int kFront = 8;                    // extra samples staged before the chunk
  static constexpr int kTot = kExt * 210 + 16;        // staged samples per channel
  static constexpr int kXBytes = C * kTot * 2;
};

// LDS map.  Region X holds the staged PCM during stage 0/1; afterwards:
//   [0, 28*kNQ)            s_bb1  float[kNQ*7]
//   then 5 x float[kNQ]    s_e, s_z, s_S0, s_Sc, s_Ss
//   then float[kNQ]        s_T0     (be2)
//   then double[kNQ]       s_R3     (bb2^2)
//   then double[7][kExt]   s_F      per-frame sums
//   then float[2*kExt]     s_eb, float[kExt] s_zf
template <int C> struct FeatLds {
  using Cfg = FeatCfg<C>;
  static constexpr int o_bb1 = 0;
  static constexpr int o_e = o_bb1 + 28 * Cfg::kNQ;
  static constexpr int o_z = o_e + 4 * Cfg::kNQ;
  static constexpr int o_S0 = o_z + 4 * Cfg::kNQ;
  static constexpr int o_Sc = o_S0 + 4 * Cfg::kNQ;
  static constexpr int o_Ss = o_Sc + 4 * Cfg::kNQ;
  static constexpr int o_T0 = o_Ss + 4 * Cfg::kNQ;
  static constexpr int o_R3 = (o_T0 + 4 * Cfg::kNQ + 7) & ~7;
  static constexpr int o_F = o_R3 + 8 * Cfg::kNQ;
  static constexpr int o_eb = o_F + 8 * 7 * Cfg::kExt;
  static constexpr int o_zf = o_eb + 4 * 2 * Cfg::kExt;
  static constexpr int o_tab = o_zf + 4 * Cfg::kExt;          // cos1/sin1 tables as float[84]
  static constexpr int aliased_end = o_tab + 4 * 84;
  static constexpr int total = ((Cfg::kXBytes > aliased_end ? Cfg::kXBytes : aliased_end) + 15) & ~15;
};

template <int C>
__global__ __launch_bounds__(FeatCfg<C>::kThreads, DA_FEAT_MIN_WAVES) void k_features(FeatArgs a, const FeatTables* __restrict__ tp) {
  using Cfg = FeatCfg<C>;
  using L = FeatLds<C>;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const uint32_t* s_xw = reinterpret_cast<const uint32_t*>(smem);           // packed halves, [C][kTot/2]
  float* s_bb1 = reinterpret_cast<float*>(smem + L::o_bb1);
  float* s_e = reinterpret_cast<float*>(smem + L::o_e);
  float* s_z = reinterpret_cast<float*>(smem + L::o_z);
  float* s_S0 = reinterpret_cast<float*>(smem + L::o_S0);
  float* s_Sc = reinterpret_cast<float*>(smem + L::o_Sc);
  float* s_Ss = reinterpret_cast<float*>(smem + L::o_Ss);
  float* s_T0 = reinterpret_cast<float*>(smem + L::o_T0);
  double* s_R3 = reinterpret_cast<double*>(smem + L::o_R3);
  double* s_F = reinterpret_cast<double*>(smem + L::o_F);
  float* s_eb = reinterpret_cast<float*>(smem + L::o_eb);
  float* s_zf = reinterpret_cast<float*>(smem + L::o_zf);

  const FeatTables& T = *tp;
  const int tid = threadIdx.x;
  const int64_t f0 = (int64_t)blockIdx.x * Cfg::kOut;            // first output frame of this chunk
  const int64_t s0 = 210 * (f0 - Cfg::kHalo) - Cfg::kFront;      // sample index of LDS slot 0

  // ---- stage 0: HBM -> LDS, int16 -> float16 (round to nearest even, as numpy astype) -------
  auto convert_store = [&](int t, const uint32_t (&w)[C][4]) {
#pragma unroll
    for (int c = 0; c < C; ++c) {
      uint32_t o[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        // both halves converted in place (sub-dword addressing): two instructions per pair of samples, no separate pack
        asm("v_cvt_f16_i16_sdwa %0, %1 dst_sel:WORD_0 dst_unused:UNUSED_PAD src0_sel:WORD_0\n\t"
            "v_cvt_f16_i16_sdwa %0, %1 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1"
            : "=&v"(o[e]) : "v"(w[c][e]));
      }
      *reinterpret_cast<uint4*>(smem + 2 * (c * Cfg::kTot + 8 * t)) = make_uint4(o[0], o[1], o[2], o[3]);
    }
  };
  constexpr int kIters = (Cfg::kTot / 8 + Cfg::kThreads - 1) / Cfg::kThreads;
  const bool interior = s0 >= 0 && s0 + Cfg::kTot <= a.n_energy && a.stride_n == 1 &&
                        ((reinterpret_cast<uintptr_t>(a.pcm) & 3) == 0) && ((a.stride_c & 1) == 0);
  if (interior) {
    // whole chunk inside the PCM (all but the first/last workgroup): issue every 16-byte load of
    // this thread first, so one HBM round trip is exposed instead of one per iteration
    uint32_t w[kIters][C][4];
#pragma unroll
    for (int it = 0; it < kIters; ++it) {
      int t = tid + it * Cfg::kThreads;
      if (t >= Cfg::kTot / 8) t = Cfg::kTot / 8 - 1;             // duplicate load, result unused
#pragma unroll
      for (int c = 0; c < C; ++c) {
        const uint4 v = *reinterpret_cast<const uint4*>(a.pcm + c * a.stride_c + s0 + 8 * (int64_t)t);
        w[it][c][0] = v.x; w[it][c][1] = v.y; w[it][c][2] = v.z; w[it][c][3] = v.w;
      }
    }
#pragma unroll
    for (int it = 0; it < kIters; ++it) {
      const int t = tid + it * Cfg::kThreads;
      if (t < Cfg::kTot / 8) convert_store(t, w[it]);
    }
  } else {
    for (int t = tid; t < Cfg::kTot / 8; t += Cfg::kThreads) {
      const int64_t n0 = s0 + 8 * (int64_t)t;
      uint32_t w[C][4];
      const bool inside = n0 >= 0 && n0 + 8 <= a.n_energy;
      if (inside && a.stride_n == 1 && ((reinterpret_cast<uintptr_t>(a.pcm + n0) & 3) == 0) && ((a.stride_c & 1) == 0)) {
#pragma unroll
        for (int c = 0; c < C; ++c) {
          const uint4 v = *reinterpret_cast<const uint4*>(a.pcm + c * a.stride_c + n0);   // dword-aligned 16-byte load
          w[c][0] = v.x; w[c][1] = v.y; w[c][2] = v.z; w[c][3] = v.w;
        }
      } else if (inside && C == 2 && a.stride_n == 2 && a.stride_c == 1) {
        const uint32_t* q = reinterpret_cast<const uint32_t*>(a.pcm + 2 * n0);   // interleaved stereo frames
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const uint32_t fa = q[2 * e], fb = q[2 * e + 1];
          w[0][e] = (fa & 0xffffu) | (fb << 16);
          w[C - 1][e] = (fa >> 16) | (fb & 0xffff0000u);
        }
      } else {
#pragma unroll
        for (int c = 0; c < C; ++c)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            uint32_t lohi[2];
#pragma unroll
            for (int b = 0; b < 2; ++b) {
              const int64_t n = n0 + 2 * e + b;
              lohi[b] = (n >= 0 && n < a.n_energy) ? (uint16_t)a.pcm[c * a.stride_c + n * a.stride_n] : 0u;
            }
            w[c][e] = lohi[0] | (lohi[1] << 16);
          }
      }
      convert_store(t, w);
    }
  }
  __syncthreads();

  // ---- stage 1: one thread per 70 samples -----------------------------------------------
  // window element t (0..79) <-> sample 70*K - 5 + t  (K = global pair index) = staged half-word
  // 70*k + 3 + t; the 41 dwords read start at half-word 70*k + 2.
  const int64_t q0g = 6 * (f0 - Cfg::kHalo);          // global group index of local group 0
  const int64_t nq_band = 6 * a.len_other;            // groups the band / zero-crossing rows may see
  float r_e[2], r_z[2], r_S0[2], r_Sc[2], r_Ss[2], r_bb[14];
  {
    const int k = tid;
    const int64_t Q0 = q0g + 2 * k;
    float m[80];
    uint32_t w[41];                     // one set of packed words for both channels (see M below)
    float e0 = 0.f, e1 = 0.f;
    int z0 = 0, z1 = 0;
    // energy and sign changes of one channel's words; own samples: half-words 6..75 of the read = dwords 3..37; group 0 = half-words 6..40
    auto energy_flips = [&](int d, uint32_t cur, uint32_t before) {
      const half2_t hv = *reinterpret_cast<const half2_t*>(&cur);
      const uint32_t flips = (cur ^ __builtin_amdgcn_alignbit(cur, before, 16)) & 0x80008000u;
      if (d < 20) {
        e0 = __builtin_amdgcn_fdot2(hv, hv, e0, false);
        z0 += __builtin_popcount(flips);
      } else if (d > 20) {
        e1 = __builtin_amdgcn_fdot2(hv, hv, e1, false);
        z1 += __builtin_popcount(flips);
      } else {
        const float lo = (float)hv[0], hi = (float)hv[1];
        e0 = fmaf(lo, lo, e0); e1 = fmaf(hi, hi, e1);
        z0 += (int)((flips >> 15) & 1u); z1 += (int)(flips >> 31);
      }
    };
    {
      const uint32_t* xp = s_xw + 35 * k + 1;
#pragma unroll
      for (int d = 0; d < 41; ++d) w[d] = xp[d];
#pragma unroll
      for (int d = 3; d <= 37; ++d) energy_flips(d, w[d], w[d - 1]);
    }
    if constexpr (C == 2) {
      // second channel: its energy / sign changes, then the band path's mono signal -- np.mean over the channels of the
      // float16 samples accumulates in float32 and rounds to float16 (:576): RN((a + b) / 2).  a / 2 and b / 2 are exact in
      // float16 (the samples are integers), so ONE packed fma per two samples, a * 0.5 + (b * 0.5), rounds the same exact
      // value once -- instead of two conversions to float32, an add, a multiply and a conversion back per sample.
      const uint32_t* xp = s_xw + (Cfg::kTot / 2) + 35 * k + 1;
      const half2_t half = {(_Float16)0.5f, (_Float16)0.5f};
      uint32_t before = 0u;
#pragma unroll
      for (int d = 0; d < 41; ++d) {
        const uint32_t u = xp[d];
        if (d >= 3 && d <= 37) energy_flips(d, u, before);
        before = u;
        const half2_t a = *reinterpret_cast<const half2_t*>(&w[d]);
        const half2_t b = *reinterpret_cast<const half2_t*>(&u);
        const half2_t mean = __builtin_elementwise_fma(a, half, b * half);
        w[d] = *reinterpret_cast<const uint32_t*>(&mean);
      }
    }
    // float32 window for the band path
#pragma unroll
    for (int t = 0; t < 80; ++t) {
      const int hw = t + 1;
      const half2_t hv = *reinterpret_cast<const half2_t*>(&w[hw >> 1]);
      m[t] = (float)hv[hw & 1];
    }
    // samples at or beyond n_band do not exist for the band rows (arr is truncated first, :577)
    const int64_t nwin0 = 35 * Q0 - 5;
    if (nwin0 + 80 > a.n_band) {
#pragma unroll
      for (int t = 0; t < 80; ++t)
        if (nwin0 + t >= a.n_band) m[t] = 0.f;
    }
    // The window is read through this accessor and w[] lives outside the channel loop: with both,
    // the register allocator keeps the stereo kernel at 176 VGPRs (two wavefronts per SIMD, 1.98
    // TB/s) instead of 256 (one per SIMD, 1.50 TB/s); the mono kernel is unaffected (163).
    auto M = [&](int t) -> float { return m[t]; };
    const float* tab = reinterpret_cast<const float*>(smem + L::o_tab);   // not yet valid: filled below
    (void)tab;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int64_t Q = Q0 + u;
      const bool in_band = (Q >= 0) && (Q < nq_band);
      const int iq = 7 * (int)(((Q % 6) + 6) % 6);
      float S0 = 0.f, Sc = 0.f, Ss = 0.f;
#pragma unroll
      for (int g = 0; g < 7; ++g) {
        const int gg = 7 * u + g;                 // group within the 70-sample item
        float bb = 0.f;
#ifdef DA_DBG_FEAT_NOFIR       // ablation (profiles/r05_features_ablation.txt): the 15-tap low-pass for free -- the ceiling of an MFMA formulation of it
        bb = M(5 + 5 * gg + 2) * T.w15[7];
#else
#pragma unroll
        for (int kk = 0; kk < 3; ++kk)
#pragma unroll
          for (int i = 0; i < 5; ++i) bb = fmaf(T.w15[i + 5 * kk], M(5 + 5 * (gg + 1 - kk) + i), bb);
#endif
        float be = 0.f;
#pragma unroll
        for (int i = 0; i < 5; ++i) {
          const float d = M(5 + 5 * gg + i) - bb;
          be = fmaf(d, d, be);
        }
        if (!in_band) { bb = 0.f; be = 0.f; }
        r_bb[gg] = bb;
        S0 += be;
        Sc = fmaf((float)T.cos1[iq + g], be, Sc);
        Ss = fmaf((float)T.sin1[iq + g], be, Ss);
      }
      r_S0[u] = S0; r_Sc[u] = Sc; r_Ss[u] = Ss;
      r_e[u] = u ? e1 : e0;
      r_z[u] = in_band ? (float)(u ? z1 : z0) : 0.f;
    }
  }
  __syncthreads();                                    // everyone is done reading the PCM region
  {
    const int k = tid;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int q = 2 * k + u;
      s_e[q] = r_e[u]; s_z[q] = r_z[u]; s_S0[q] = r_S0[u]; s_Sc[q] = r_Sc[u]; s_Ss[q] = r_Ss[u];
#pragma unroll
      for (int g = 0; g < 7; ++g) s_bb1[7 * q + g] = r_bb[7 * u + g];
    }
  }
  __syncthreads();

  // ---- stage 2: second low-pass level ---------------------------------------------------
  for (int q = tid; q < Cfg::kNQ; q += Cfg::kThreads) {
    const int64_t Q = q0g + q;
    float be2 = 0.f;
    double r3 = 0.0;
    if (q >= 1 && q < Cfg::kNQ - 1 && Q >= 0 && Q < nq_band) {
      const float* bp = s_bb1 + 7 * (q - 1);
      float bb2 = 0.f;
#ifdef DA_DBG_FEAT_NOFIR
      bb2 = bp[7 + 3] * T.w21[10];
#else
#pragma unroll
      for (int kk = 0; kk < 3; ++kk)
#pragma unroll
        for (int i = 0; i < 7; ++i) bb2 = fmaf(T.w21[i + 7 * kk], bp[7 * (2 - kk) + i], bb2);
#endif
#pragma unroll
      for (int i = 0; i < 7; ++i) {
        const float d = bp[7 + i] - bb2;
        be2 = fmaf(d, d, be2);
      }
      r3 = (double)bb2 * (double)bb2;     // band 3 is float64 in the reference (:583, :588)
    }
    s_T0[q] = be2;
    s_R3[q] = r3;
  }
  __syncthreads();

  // ---- stage 3: per frame -------------------------------------------------------------------
  for (int g = tid; g < Cfg::kExt; g += Cfg::kThreads) {
    double S0 = 0, Sc = 0, Ss = 0, T0 = 0, Tc = 0, Ts = 0, R = 0;
    float z = 0.f;
#pragma unroll
    for (int s = 0; s < 6; ++s) {
      const int q = 6 * g + s;
      S0 += (double)s_S0[q]; Sc += (double)s_Sc[q]; Ss += (double)s_Ss[q];
      const double b2 = (double)s_T0[q];
      T0 += b2; Tc = fma(T.cos2[s], b2, Tc); Ts = fma(T.sin2[s], b2, Ts);
      R += s_R3[q];
      z += s_z[q];
    }
    s_F[0 * Cfg::kExt + g] = S0; s_F[1 * Cfg::kExt + g] = Sc; s_F[2 * Cfg::kExt + g] = Ss;
    s_F[3 * Cfg::kExt + g] = T0; s_F[4 * Cfg::kExt + g] = Tc; s_F[5 * Cfg::kExt + g] = Ts;
    s_F[6 * Cfg::kExt + g] = R;
    s_zf[g] = (C == 1) ? 2.f * z : z;                                      // (:561-562)
    const float inv = 1.0f / (float)(105 * C);                             // (:550)
    s_eb[2 * g] = (s_e[6 * g] + s_e[6 * g + 1] + s_e[6 * g + 2]) * inv;
    s_eb[2 * g + 1] = (s_e[6 * g + 3] + s_e[6 * g + 4] + s_e[6 * g + 5]) * inv;
  }
  __syncthreads();

  // ---- stage 4: outputs -----------------------------------------------------------------------
  // One lane per (row, output frame), the rows dealt over the workgroup's three waves: the float64-logarithm
  // row on wave 0, the two float64 band rows on waves 1 and 2 together with the two light rows.  (With all five
  // rows of a frame on one lane -- the first version -- wave 0 worked alone through ~500 dependent, mostly
  // float64 instructions while the other two waves had already finished and the workgroup's slot stayed taken:
  // measured on the streaming experiment, DESIGN.md section 4.1, this stage was a third of the kernel.)
  {
    const int wave = tid >> 6, ml = tid & 63;
    const int64_t mfr = f0 + ml;
    const int g = ml + Cfg::kHalo;
    if (ml < Cfg::kOut && wave < 3) {
      if (wave == 0) {
        if (mfr < a.len_other) {
          double b3 = 0.0;
#pragma unroll
          for (int k = 0; k < 15; ++k) b3 = fma((double)T.w15[k], s_F[6 * Cfg::kExt + g + 7 - k], b3);
          a.out[4 * a.row_stride + mfr] = (float)(log10(1.0 + b3 / 210.0) * 0.5);
        }
      } else if (wave == 1) {
        if (mfr < a.len_other) {
          double b1 = 0.0;
#pragma unroll
          for (int k = 0; k < 15; ++k) {
            const int gg = g + 7 - k;
            b1 += s_F[0 * Cfg::kExt + gg] - T.ck1[k] * s_F[1 * Cfg::kExt + gg] + T.sk1[k] * s_F[2 * Cfg::kExt + gg];
          }
          b1 *= T.a1;
          a.out[2 * a.row_stride + mfr] = log10f(1.0f + (float)(b1 / 210.0)) * 0.5f;   // (:589-590)
        }
        if (mfr < a.len_energy) {
          float acc = 0.f;
#pragma unroll
          for (int t = -6; t <= 6; ++t) acc = fmaf(T.w13[t + 6], s_eb[2 * g + t], acc);
          a.out[0 * a.row_stride + mfr] = log10f(1.0f + acc) * 0.5f;           // (:554)
        }
      } else {
        if (mfr < a.len_other) {
          double b2 = 0.0;
#pragma unroll
          for (int k = 0; k < 15; ++k) {
            const int gg = g + 7 - k;
            b2 += s_F[3 * Cfg::kExt + gg] - T.ck2[k] * s_F[4 * Cfg::kExt + gg] + T.sk2[k] * s_F[5 * Cfg::kExt + gg];
          }
          b2 *= T.a2;
          a.out[3 * a.row_stride + mfr] = log10f(1.0f + (float)(b2 / 210.0)) * 0.5f;
          float zacc = 0.f;
#pragma unroll
          for (int t = -6; t <= 6; ++t) zacc = fmaf(T.w13[t + 6], s_zf[g + t], zacc);
          a.out[1 * a.row_stride + mfr] = zacc;
        }
      }
    }
  }
  static_assert(Cfg::kThreads >= 192 && Cfg::kOut <= 64, "stage 4 deals the rows over three waves");
}

template <int C> static void launch_one(const FeatArgs& a, const FeatTables* d_tables, hipStream_t s) {
  using Cfg = FeatCfg<C>;
  const int64_t blocks = (a.len_energy + Cfg::kOut - 1) / Cfg::kOut;
  const int smem = FeatLds<C>::total;
  // per launch: the attribute is per device, and contexts on several devices / threads share this code
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_features<C>), hipFuncAttributeMaxDynamicSharedMemorySize, smem);
  hipLaunchKernelGGL(k_features<C>, dim3((unsigned)blocks), dim3(Cfg::kThreads), smem, s, a, d_tables);
}

void launch_features(const FeatArgs& a, int channels, const FeatTables* d_tables, hipStream_t s) {
  if (a.len_energy <= 0) return;
  if (channels == 1) launch_one<1>(a, d_tables, s);
  else launch_one<2>(a, d_tables, s);
}

}  // namespace da
