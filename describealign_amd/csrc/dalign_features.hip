// Fused feature kernel: int16 PCM -> 5 feature rows at 210 frames/s (gfx950).
//
// Replaces get_energy / get_zero_crossings / downsample_blur / get_freq_bands
// (describealign.py:545-593).  One pass over the PCM: a workgroup stages a chunk of frames
// (plus an 8-frame halo either side) from HBM into LDS with 16-byte loads, rounding through
// float16 exactly as the reference's array does (:156), and everything downstream runs out of
// LDS:
//   stage 1  one thread per 35-sample group: block energy and sign-change partials, the 5x3
//            low-pass (bb1), band-1 residual energy (be1) and its three separable blur sums
//   stage 2  one thread per 35-sample group: the 7x3 low-pass (bb2), band-2 residual energy
//            and its separable blur sums, band-3 energy
//   stage 3  one thread per frame: gather the six groups of the frame
//   stage 4  one thread per output frame: 13-tap Hann smoothing of energy / zero crossings,
//            15-tap combination of the blur sums, log10(1+x)/2, store.
// The 630-tap and 90-tap Hann blurs of the reference are evaluated in their exactly
// equivalent separable form (see FeatTables) in float64, so no precision is lost by it.
//
// Roofline: HBM bound.  Algorithmic bytes = 2*C*N read + 5 rows * 4 B * N/210 written.
#include "dalign_common.h"

namespace da {

template <int C> struct FeatCfg {
  static constexpr int kExt = (C == 1) ? 128 : 64;    // frames staged per workgroup (incl. halo)
  static constexpr int kHalo = 8;
  static constexpr int kOut = kExt - 2 * kHalo;       // frames produced per workgroup
  static constexpr int kNQ = kExt * 6;                // 35-sample groups per chunk
  static constexpr int kFront = 8;                    // extra samples staged before the chunk
  static constexpr int kTot = kExt * 210 + 16;        // staged samples per channel
};

constexpr int kThreads = 256;

template <int C>
__global__ __launch_bounds__(kThreads) void k_features(FeatArgs a, const FeatTables* __restrict__ tp) {
  using Cfg = FeatCfg<C>;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  _Float16* s_x = reinterpret_cast<_Float16*>(smem);                        // [C][kTot]
  double* s_d = reinterpret_cast<double*>(smem + ((C * Cfg::kTot * 2 + 15) & ~15));
  double* s_S0 = s_d;                    // [kNQ] band-1 blur partials per group
  double* s_Sc = s_S0 + Cfg::kNQ;
  double* s_Ss = s_Sc + Cfg::kNQ;
  double* s_T0 = s_Ss + Cfg::kNQ;        // [kNQ] band-2: be2, and bb2^2
  double* s_R3 = s_T0 + Cfg::kNQ;
  double* s_F = s_R3 + Cfg::kNQ;         // [7][kExt] per-frame: S0,Sc,Ss,T0,Tc,Ts,be3
  float* s_bb1 = reinterpret_cast<float*>(s_F + 7 * Cfg::kExt);   // [kNQ*7]
  float* s_e = s_bb1 + Cfg::kNQ * 7;     // [kNQ] energy partial (sum of squares over 35 samples, all channels)
  float* s_z = s_e + Cfg::kNQ;           // [kNQ] sign-change partial
  float* s_eb = s_z + Cfg::kNQ;          // [2*kExt] block energies
  float* s_zf = s_eb + 2 * Cfg::kExt;    // [kExt] frame sign changes

  const FeatTables& T = *tp;
  const int tid = threadIdx.x;
  const int64_t f0 = (int64_t)blockIdx.x * Cfg::kOut;            // first output frame of this chunk
  const int64_t s0 = 210 * (f0 - Cfg::kHalo) - Cfg::kFront;      // sample index of LDS slot 0

  // ---- stage 0: HBM -> LDS, int16 -> float16 (round to nearest even, as numpy astype) -------
  for (int t = tid; t < Cfg::kTot / 8; t += kThreads) {
    const int64_t n0 = s0 + 8 * (int64_t)t;
    short v[C][8];
    if (n0 >= 0 && n0 + 8 <= a.n_energy && a.stride_n == 1) {
#pragma unroll
      for (int c = 0; c < C; ++c) {
        const int16_t* p = a.pcm + c * a.stride_c + n0;
        // 4-byte aligned 16-byte load (n0 is even whenever the base is 4-byte aligned)
        if ((reinterpret_cast<uintptr_t>(p) & 3) == 0) {
          const uint32_t* q = reinterpret_cast<const uint32_t*>(p);
          uint32_t w0 = q[0], w1 = q[1], w2 = q[2], w3 = q[3];
          v[c][0] = (short)(w0 & 0xffff); v[c][1] = (short)(w0 >> 16);
          v[c][2] = (short)(w1 & 0xffff); v[c][3] = (short)(w1 >> 16);
          v[c][4] = (short)(w2 & 0xffff); v[c][5] = (short)(w2 >> 16);
          v[c][6] = (short)(w3 & 0xffff); v[c][7] = (short)(w3 >> 16);
        } else {
#pragma unroll
          for (int e = 0; e < 8; ++e) v[c][e] = p[e];
        }
      }
    } else if (n0 >= 0 && n0 + 8 <= a.n_energy && C == 2 && a.stride_n == 2 && a.stride_c == 1) {
      // interleaved stereo: 8 frames = 32 bytes
      const uint32_t* q = reinterpret_cast<const uint32_t*>(a.pcm + 2 * n0);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        uint32_t w = q[e];
        v[0][e] = (short)(w & 0xffff);
        if (C == 2) v[C - 1][e] = (short)(w >> 16);
      }
    } else {
#pragma unroll
      for (int c = 0; c < C; ++c)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const int64_t n = n0 + e;
          v[c][e] = (n >= 0 && n < a.n_energy) ? a.pcm[c * a.stride_c + n * a.stride_n] : (short)0;
        }
    }
#pragma unroll
    for (int c = 0; c < C; ++c) {
      _Float16 h[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) h[e] = (_Float16)(float)v[c][e];
      *reinterpret_cast<uint4*>(&s_x[c * Cfg::kTot + 8 * t]) = *reinterpret_cast<const uint4*>(h);
    }
  }
  __syncthreads();

  // ---- stage 1: per 35-sample group --------------------------------------------------------
  const int64_t q0 = 6 * (f0 - Cfg::kHalo);          // global group index of local group 0
  const int64_t nq_band = 6 * a.len_other;           // groups the band / zero-crossing rows may see
  for (int q = tid; q < Cfg::kNQ; q += kThreads) {
    const int64_t Q = q0 + q;
    const int base = Cfg::kFront + 35 * q - 5;       // LDS slot of window element 0
    float m[45];
    float esum = 0.f;
    int zc = 0;
#pragma unroll
    for (int c = 0; c < C; ++c) {
      const _Float16* xp = s_x + c * Cfg::kTot + base;
      float x[45];
#pragma unroll
      for (int t = 0; t < 45; ++t) x[t] = (float)xp[t];
#pragma unroll
      for (int t = 5; t < 40; ++t) {
        esum = fmaf(x[t], x[t], esum);
        zc += (int)((__float_as_uint(x[t]) ^ __float_as_uint(x[t - 1])) >> 31);
      }
      if (c == 0) {
#pragma unroll
        for (int t = 0; t < 45; ++t) m[t] = x[t];
      } else {
        // channel mean kept in float16, accumulated in float32 (np.mean on a float16 array, :576)
#pragma unroll
        for (int t = 0; t < 45; ++t) m[t] = (float)(_Float16)((m[t] + x[t]) * 0.5f);
      }
    }
    const bool in_band = (Q >= 0) && (Q < nq_band);
    // samples at or beyond n_band do not exist for the band rows (arr is truncated first, :577)
    const int64_t nwin0 = 35 * Q - 5;
    if (nwin0 + 45 > a.n_band) {
#pragma unroll
      for (int t = 0; t < 45; ++t)
        if (nwin0 + t >= a.n_band) m[t] = 0.f;
    }
    s_e[q] = esum;
    s_z[q] = in_band ? (float)zc : 0.f;
    double S0 = 0.0, Sc = 0.0, Ss = 0.0;
    const int iq = 7 * (int)(((Q % 6) + 6) % 6);
#pragma unroll
    for (int g = 0; g < 7; ++g) {
      float bb = 0.f;
#pragma unroll
      for (int k = 0; k < 3; ++k)
#pragma unroll
        for (int i = 0; i < 5; ++i) bb = fmaf(T.w15[i + 5 * k], m[5 + 5 * (g + 1 - k) + i], bb);
      float be = 0.f;
#pragma unroll
      for (int i = 0; i < 5; ++i) {
        const float d = m[5 + 5 * g + i] - bb;
        be = fmaf(d, d, be);
      }
      if (!in_band) { bb = 0.f; be = 0.f; }
      s_bb1[7 * q + g] = bb;
      const double bed = (double)be;
      S0 += bed;
      Sc = fma(T.cos1[iq + g], bed, Sc);
      Ss = fma(T.sin1[iq + g], bed, Ss);
    }
    s_S0[q] = S0; s_Sc[q] = Sc; s_Ss[q] = Ss;
  }
  __syncthreads();

  // ---- stage 2: second low-pass level ---------------------------------------------------
  for (int q = tid; q < Cfg::kNQ; q += kThreads) {
    const int64_t Q = q0 + q;
    double be2 = 0.0, r3 = 0.0;
    if (q >= 1 && q < Cfg::kNQ - 1 && Q >= 0 && Q < nq_band) {
      const float* bp = s_bb1 + 7 * (q - 1);
      float bb2 = 0.f;
#pragma unroll
      for (int k = 0; k < 3; ++k)
#pragma unroll
        for (int i = 0; i < 7; ++i) bb2 = fmaf(T.w21[i + 7 * k], bp[7 * (2 - k) + i], bb2);
      float be = 0.f;
#pragma unroll
      for (int i = 0; i < 7; ++i) {
        const float d = bp[7 + i] - bb2;
        be = fmaf(d, d, be);
      }
      be2 = (double)be;
      r3 = (double)bb2 * (double)bb2;     // band 3 is float64 in the reference (:583, :588)
    }
    s_T0[q] = be2;
    s_R3[q] = r3;
  }
  __syncthreads();

  // ---- stage 3: per frame -------------------------------------------------------------------
  for (int g = tid; g < Cfg::kExt; g += kThreads) {
    double S0 = 0, Sc = 0, Ss = 0, T0 = 0, Tc = 0, Ts = 0, R = 0;
    float z = 0.f;
#pragma unroll
    for (int s = 0; s < 6; ++s) {
      const int q = 6 * g + s;
      S0 += s_S0[q]; Sc += s_Sc[q]; Ss += s_Ss[q];
      const double b2 = s_T0[q];
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
  for (int ml = tid; ml < Cfg::kOut; ml += kThreads) {
    const int64_t mfr = f0 + ml;
    const int g = ml + Cfg::kHalo;
    if (mfr < a.len_energy) {
      float acc = 0.f;
#pragma unroll
      for (int t = -6; t <= 6; ++t) acc = fmaf(T.w13[t + 6], s_eb[2 * g + t], acc);
      a.out[0 * a.row_stride + mfr] = log10f(1.0f + acc) * 0.5f;           // (:554)
    }
    if (mfr < a.len_other) {
      float zacc = 0.f;
#pragma unroll
      for (int t = -6; t <= 6; ++t) zacc = fmaf(T.w13[t + 6], s_zf[g + t], zacc);
      a.out[1 * a.row_stride + mfr] = zacc;
      double b1 = 0.0, b2 = 0.0, b3 = 0.0;
#pragma unroll
      for (int k = 0; k < 15; ++k) {
        const int gg = g + 7 - k;
        b1 += s_F[0 * Cfg::kExt + gg] - T.ck1[k] * s_F[1 * Cfg::kExt + gg] + T.sk1[k] * s_F[2 * Cfg::kExt + gg];
        b2 += s_F[3 * Cfg::kExt + gg] - T.ck2[k] * s_F[4 * Cfg::kExt + gg] + T.sk2[k] * s_F[5 * Cfg::kExt + gg];
        b3 = fma((double)T.w15[k], s_F[6 * Cfg::kExt + gg], b3);
      }
      b1 *= T.a1; b2 *= T.a2;
      a.out[2 * a.row_stride + mfr] = log10f(1.0f + (float)(b1 / 210.0)) * 0.5f;   // (:589-590)
      a.out[3 * a.row_stride + mfr] = log10f(1.0f + (float)(b2 / 210.0)) * 0.5f;
      a.out[4 * a.row_stride + mfr] = (float)(log10(1.0 + b3 / 210.0) * 0.5);
    }
  }
}

template <int C> static size_t feat_smem_bytes() {
  using Cfg = FeatCfg<C>;
  size_t b = (C * Cfg::kTot * 2 + 15) & ~size_t(15);
  b += sizeof(double) * (5 * Cfg::kNQ + 7 * Cfg::kExt);
  b += sizeof(float) * (Cfg::kNQ * 7 + 2 * Cfg::kNQ + 3 * Cfg::kExt);
  return b;
}

void launch_features(const FeatArgs& a, int channels, const FeatTables* d_tables, hipStream_t s) {
  if (a.len_energy <= 0) return;
  if (channels == 1) {
    const int64_t blocks = (a.len_energy + FeatCfg<1>::kOut - 1) / FeatCfg<1>::kOut;
    static bool once = false;
    if (!once) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_features<1>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)feat_smem_bytes<1>());
      once = true;
    }
    hipLaunchKernelGGL(k_features<1>, dim3((unsigned)blocks), dim3(kThreads), feat_smem_bytes<1>(), s, a, d_tables);
  } else {
    const int64_t blocks = (a.len_energy + FeatCfg<2>::kOut - 1) / FeatCfg<2>::kOut;
    static bool once = false;
    if (!once) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_features<2>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)feat_smem_bytes<2>());
      once = true;
    }
    hipLaunchKernelGGL(k_features<2>, dim3((unsigned)blocks), dim3(kThreads), feat_smem_bytes<2>(), s, a, d_tables);
  }
}

}  // namespace da
