// Matching kernels (gfx950): per-side preparation, the similarity GEMM on MFMA, exact
// verification of the GEMM's survivors, and the pass-2 banded line evaluation.
//
// The reference scores a candidate (audio frame i, video frame v) with three 41-tap windowed
// correlations  corr_j = <A_j[i:i+41], V_j[v:v+41]> / (|A_j|_i |V_j|_v)  and keeps it when
// (prod_j max(1e-8, 1 - corr_j))^2.9 <= 1e-8  (describealign.py:662-671).  Over all pairs that
// is a Hankel GEMM with K = 41 per feature.  k_match_* evaluates it densely on the matrix cores
// for (non-quiet audio frames) x (every 4th non-quiet video frame), thresholds in the epilogue
// and appends the rare survivors to a list; k_verify then recomputes those pairs exactly in
// float64, applies the reference's hash vote (:649-660) in its closed form, and emits
// (i, v, quality).
#include "dalign_common.h"
#include <cstdlib>
#include <string>

namespace da {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));   // 16-byte load, 4-byte aligned
typedef uint32_t u32x4u __attribute__((ext_vector_type(4), aligned(4)));

// ------------------------------------------------------------------------------------------
// prep: mean subtraction (:598-599, :605-606), window norms (:600-602), hash digits (:623-628,
// :639-643) and the GEMM operand copies.  One thread per frame.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_prep_ms(PrepArgs a, const double* __restrict__ w /*41, normalised*/) {
  const int j = blockIdx.y;
  const int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t L = a.len[j];
  if (n >= a.lmax + kPad) return;
  double out = 0.0;
  if (n < L) {
    const float* f = a.feat + (int64_t)j * a.row_stride;
    double acc = 0.0;
#pragma unroll
    for (int t = -20; t <= 20; ++t) {
      const int64_t m = n + t;
      const double x = (m >= 0 && m < L) ? (double)f[m] : 0.0;
      acc += w[20 + t] * x;        // same tap order as a direct convolution
    }
    out = (double)f[n] - acc;
  }
  a.ms[j][n] = out;
  if (j < 3) {
    a.ms32[j][n] = (float)out;
  }
}

__device__ inline uint16_t f32_to_bf16(float x) {
  __bf16 b = (__bf16)x;                     // v_cvt_pk_bf16_f32 (round to nearest even)
  return *reinterpret_cast<uint16_t*>(&b);
}

__global__ __launch_bounds__(256) void k_prep_norm(PrepArgs a) {
  const int j = blockIdx.y;
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t L = a.len[j];
  if (i >= a.lmax + kPad) return;
  const double* ms = a.ms[j];
  if (j < 3) {
    // bf16 copies of the (unnormalised) mean-subtracted row, scaled into range by nothing: values are O(1)
    a.bf_even[j][i] = f32_to_bf16((float)ms[i]);
    a.bf_odd[j][i] = f32_to_bf16((i + 1 < a.lmax + kPad) ? (float)ms[i + 1] : 0.f);
  }
  const int64_t nv = L - (kWin - 1);
  if (i >= nv || nv <= 0) {
    if (i < a.lmax + kPad) {
      a.nrm[j][i] = 1.0;
      a.digits[j][i] = 0xFFFFFFFFu;          // never matches
      if (a.is_video) a.flags[j][i] = 0xFFFFFFFFu;
      if (j < 3) { a.inv32[j][i] = 0.f; a.nrm32[j][i] = 1.f; a.nrmpk[j][i] = 0x00003F80u; }
    }
    return;
  }
  double ss = 0.0;
#pragma unroll
  for (int k = 0; k < kWin; ++k) ss += ms[i + k] * ms[i + k];
  double nr = sqrt(ss);
  nr = nr < 0.001 ? 0.001 : nr;
  a.nrm[j][i] = nr;
  if (j < 3) {
    a.inv32[j][i] = (float)(1.0 / nr); a.nrm32[j][i] = (float)nr;
    const uint16_t hi = f32_to_bf16((float)nr);
    const float hif = __uint_as_float((uint32_t)hi << 16);
    const uint16_t lo = f32_to_bf16((float)(nr - (double)hif));
    a.nrmpk[j][i] = (uint32_t)hi | ((uint32_t)lo << 16);
  }
  uint32_t dig = 0, flg = 0;
#pragma unroll
  for (int b = 0; b < kTaps; ++b) {
    const double tap = ms[i + kTapStart + kTapStep * b] / nr;
    if (a.is_video) {
      double u = 8.0 * tap + 3.3;
      u = u < 0.0 ? 0.0 : (u > 6.0 ? 6.0 : u);
      const double fl = floor(u);
      dig |= (uint32_t)(int)fl << (4 * b);
      if (u - fl > 0.6) flg |= 1u << (4 * b);
    } else {
      double u = floor(8.0 * tap + 3.5);
      u = u < 0.0 ? 0.0 : (u > 6.0 ? 6.0 : u);
      dig |= (uint32_t)(int)u << (4 * b);
    }
  }
  if (a.is_video) {
    a.digits[j][i] = dig;
    a.flags[j][i] = ~flg;
  } else {
    a.digits[j][i] = dig | 0x08888888u;      // guard bit per nibble: no borrows in the packed subtract
  }
}

__global__ __launch_bounds__(256) void k_prep_prod(PrepArgs a) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= a.lmax + kPad) return;
  a.prod32[i] = (float)(a.nrm[0][i] * a.nrm[1][i] * a.nrm[2][i]);
}

void launch_prep(const PrepArgs& a, const double* d_hann41n, hipStream_t s) {
  const int64_t n = a.lmax + kPad;
  dim3 grid((unsigned)((n + 255) / 256), 5);
  hipLaunchKernelGGL(k_prep_ms, grid, dim3(256), 0, s, a, d_hann41n);
  hipLaunchKernelGGL(k_prep_norm, grid, dim3(256), 0, s, a);
  hipLaunchKernelGGL(k_prep_prod, dim3(grid.x), dim3(256), 0, s, a);
}

// hash vote in closed form: feature j "hits" when every audio digit equals
// the video digit, or the video digit + 1 where the video flag is set.
__device__ inline bool digit_hit(uint32_t a_guarded, uint32_t v_dig, uint32_t v_notflag) {
  const uint32_t d = ((a_guarded - v_dig) ^ 0x08888888u);   // per nibble: 0 equal, 1 one above
  return (d & v_notflag & 0x0FFFFFFFu) == 0u && ((d & 0x0EEEEEEEu) == 0u);
}

// ------------------------------------------------------------------------------------------
// similarity GEMM, float32 inputs on v_mfma_f32_32x32x2_f32
//
// Tile: one wave owns 32 video rows (MFMA A operand, held in 63 VGPRs for the whole launch,
// pre-scaled by -1/|V|) and streams 32 audio columns at a time (MFMA B operand, straight from
// L1/L2).  The K index is permuted (lane half h, step s -> k = 21 h + s; k = 41 is the zero pad)
// so that each lane's operands are 21 consecutive floats.
// acc = -<A,V>/|V|;  t_j = 1 + acc * (1/|A|_i) = 1 - corr_j;  survivor when t_0 t_1 t_2 <= thr.
// ------------------------------------------------------------------------------------------
constexpr int kWavesPerBlock = 4;
constexpr int kSurvBuf = 256;          // survivor staging slots per wave (LDS)

struct SurvSink {
  unsigned long long* s_buf;           // this wave's LDS staging
  int count;                           // wave-uniform
  int cap = kSurvBuf;                  // slots in s_buf
};

__device__ inline void sink_flush(SurvSink& sk, const MatchArgs& a, int lane) {
  if (sk.count == 0) return;
  unsigned long long base = 0;
  if (lane == 0) base = atomicAdd(a.out_count, (unsigned long long)sk.count);
  base = __shfl(base, 0);
  for (int t = lane; t < sk.count; t += 64) {
    const unsigned long long pos = base + t;
    if (pos < a.capacity) a.out[pos] = sk.s_buf[t];
  }
  sk.count = 0;
}

__device__ inline void sink_push(SurvSink& sk, const MatchArgs& a, int lane, bool pass, unsigned long long rec) {
  const unsigned long long m = __ballot(pass);
  if (m == 0ull) return;
  const int n = __popcll(m);
  if (sk.count + n > sk.cap) sink_flush(sk, a, lane);
  if (pass) {
    const int pos = sk.count + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
    sk.s_buf[pos] = rec;
  }
  sk.count += n;
}

// Software pipeline (one wave per SIMD, 512-VGPR budget): while the 63 MFMAs of tile t run, the
// operand loads of tile t+1 are in flight and the threshold epilogue of tile t-1 is interleaved
// between the MFMAs (VALU and matrix pipes overlap).  Two register sets are ping-ponged by
// unrolling the tile loop by two, so nothing is copied.
struct TileMeta {
  float thr;      // threshold of this lane's audio column: thr * |A|_0 |A|_1 |A|_2
  int32_t ic;
  bool ok;
};

// audio frame number of column r of the tile starting at list position `at` (clamped at the end)
__device__ __forceinline__ int32_t fetch_index(const MatchArgs& a, int64_t at, int64_t a_end, int r) {
  int64_t ia = at + r;
  if (ia >= a_end) ia = a_end - 1;
  return a.alist[ia];
}

// Issue the operand loads of the tile at list position `at`, whose frame numbers `ic` were fetched
// one phase earlier (a dependent index->operand load chain at the head of every phase would
// expose a full memory round trip with no MFMA in flight).
__device__ __forceinline__ void load_tile_b(const MatchArgs& a, int64_t at, int64_t a_end, int r, int h, int32_t ic,
                                            float (&b)[3][21], TileMeta& m) {
  m.ok = (at + r) < a_end;
  m.ic = ic;
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    // K permutation: lanes with h = 0 hold k = s, lanes with h = 1 hold k = 21 + s, so a lane's 21
    // operands are contiguous: 5 x 16-byte + 1 x 4-byte loads (4-byte aligned) instead of 21 dword
    // loads -- a wave may only have 63 vector-memory operations in flight (6-bit vmcnt).
    const float* p = a.ms_a[j] + m.ic + 21 * h;
#pragma unroll
    for (int q = 0; q < 5; ++q) {
      const f32x4u w = *reinterpret_cast<const f32x4u*>(p + 4 * q);
      b[j][4 * q + 0] = w[0]; b[j][4 * q + 1] = w[1]; b[j][4 * q + 2] = w[2]; b[j][4 * q + 3] = w[3];
    }
    // last K slot: k = 20 for h = 0; for h = 1 it is the spare slot k = 41, which carries the audio
    // window norm (the A operand holds 1 there), so the accumulator ends as |A| - <A,V>/|V| = |A| (1 - corr)
    b[j][20] = h ? a.nrm_a[j][m.ic] : p[20];
  }
  m.thr = a.thr * a.prod_a[m.ic];
}

// acc_j = |A|_j (1 - corr_j); survivor when prod_j acc_j <= thr |A|_0 |A|_1 |A|_2 (same test as
// prod_j (1 - corr_j) <= thr, with the per-column norms folded into the threshold)
__device__ __forceinline__ uint32_t threshold_row(const f32x16 (&acc)[3], const TileMeta& m, int g, float) {
  return (acc[0][g] * acc[1][g] * acc[2][g] <= m.thr) ? (1u << g) : 0u;
}

// Two rows at once: packed products, then for each row  mask = 2*mask + (prod <= thr)  as one
// compare + one add-with-carry.  Rows must be fed in descending order (row g ends at bit g).
__device__ __forceinline__ void threshold_rows2(const f32x16 (&acc)[3], float thr, int g_hi, uint32_t& mask) {
  const f32x2 x0 = {acc[0][g_hi - 1], acc[0][g_hi]};
  const f32x2 x1 = {acc[1][g_hi - 1], acc[1][g_hi]};
  const f32x2 x2 = {acc[2][g_hi - 1], acc[2][g_hi]};
  // two packed multiplies for the two rows (the compiler scalarises the vector expression: the
  // operands come out of AGPRs, so pairing them costs nothing when asked for explicitly)
  f32x2 pr;
  asm("v_pk_mul_f32 %0, %1, %2" : "=v"(pr) : "v"(x0), "v"(x1));
  asm("v_pk_mul_f32 %0, %1, %2" : "=v"(pr) : "v"(pr), "v"(x2));
  asm("v_cmp_le_f32 vcc, %1, %2\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(mask) : "v"(pr[1]), "v"(thr) : "vcc");
  asm("v_cmp_le_f32 vcc, %1, %2\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(mask) : "v"(pr[0]), "v"(thr) : "vcc");
}

// Survivors of one tile: mask bit g <-> accumulator register g of this lane.  Each lane that has
// any stages ONE compact record  [63:41] audio frame | [40:17] video tile | [16] lane half | [15:0] mask
// (wave-aggregated slot in LDS, no loops here); k_verify expands the bits into (i, v) pairs.
__device__ __forceinline__ unsigned long long pack_record(int32_t ic, int64_t vtile, int h, uint32_t mask) {
  return ((unsigned long long)(uint32_t)ic << 41) | ((unsigned long long)vtile << 17) | ((unsigned long long)h << 16) | mask;
}

__device__ __forceinline__ void emit_tile(SurvSink& sk, const MatchArgs& a, int lane, int h, int64_t vtile, uint32_t mask, int32_t ic) {
  sink_push(sk, a, lane, mask != 0u, pack_record(ic, vtile, h, mask));
}

// MFMAs of the current tile into acc, with the threshold epilogue of the PREVIOUS tile (accp/prev)
// interleaved between them (VALU issues under the matrix pipe), and the previous tile's survivors
// emitted while the last MFMAs are still in the pipe.
__device__ __forceinline__ void mfma_tile_f32(const float (&A)[3][21], const float (&b)[3][21], f32x16 (&acc)[3],
                                              const f32x16 (&accp)[3], const TileMeta& prev, float thr,
                                              SurvSink& sk, const MatchArgs& a, int lane, int h, int64_t vtile) {
  uint32_t mask = 0;
#pragma unroll
  for (int s = 0; s < 17; ++s) {
    acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(A[0][s], b[0][s], acc[0], 0, 0, 0);
    acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(A[1][s], b[1][s], acc[1], 0, 0, 0);
    acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(A[2][s], b[2][s], acc[2], 0, 0, 0);
#ifndef DA_DBG_NO_EPILOGUE
    if (s < 16 && (s & 1) == 0) threshold_rows2(accp, prev.thr, 15 - s, mask);     // rows 15-s, 14-s
#endif
    // per MFMA triple: one operand load of the next tile and a slice of the epilogue
    __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);
    __builtin_amdgcn_sched_group_barrier(0x020, 2, 0);
    __builtin_amdgcn_sched_group_barrier(0x002, 12, 0);
  }
#ifdef DA_DBG_NO_EPILOGUE
  asm volatile("" ::"v"(accp[0][0]), "v"(accp[1][5]), "v"(accp[2][15]));
#endif
#ifdef DA_DBG_NO_EMIT
  asm volatile("" ::"v"(mask));
#else
  emit_tile(sk, a, lane, h, vtile, prev.ok ? mask : 0u, prev.ic);
#endif
#pragma unroll
  for (int s = 17; s < 21; ++s) {
    acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(A[0][s], b[0][s], acc[0], 0, 0, 0);
    acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(A[1][s], b[1][s], acc[1], 0, 0, 0);
    acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(A[2][s], b[2][s], acc[2], 0, 0, 0);
  }
}

// All operand loads of the tile about to be consumed were issued one phase ago; naming them here
// makes the compiler retire them (vmcnt) BEFORE the next prefetch batch is issued.  Otherwise its
// in-order vmcnt bookkeeping (6-bit counter) would make the waits for these operands also wait
// for part of the fresh batch.
__device__ __forceinline__ void retire_loads(const float (&b)[3][21], const TileMeta& m) {
#pragma unroll
  for (int j = 0; j < 3; ++j) {
#pragma unroll
    for (int s = 0; s < 21; ++s) asm volatile("" ::"v"(b[j][s]));
  }
  asm volatile("" ::"v"(m.thr));
}

__global__ __launch_bounds__(64 * kWavesPerBlock, 1) void k_match_f32(MatchArgs a) {
  __shared__ unsigned long long s_surv[kWavesPerBlock][kSurvBuf];
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int64_t vt0 = ((int64_t)blockIdx.x * kWavesPerBlock + wave) * 32;
  SurvSink sk{s_surv[wave], 0};
  const int64_t a_begin = (int64_t)blockIdx.y * a.audio_tiles_per_block * 32;
  int64_t a_end = a_begin + (int64_t)a.audio_tiles_per_block * 32;
  if (a_end > a.n_a) a_end = a.n_a;
  if (vt0 < a.n_v && a_begin < a_end) {
    // fixed operand: 32 video rows, pre-scaled by -1/|V|
    const int64_t vr = vt0 + r;
    const bool vok = vr < a.n_v;
    const int32_t v = a.vlist[vok ? vr : a.n_v - 1];
    float A[3][21];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const float sc = vok ? -a.inv_v[j][v] : 0.f;
      const float* p = a.ms_v[j] + v + 21 * h;
#pragma unroll
      for (int s = 0; s < 21; ++s) A[j][s] = (21 * h + s < kWin) ? p[s] * sc : 1.0f;    // k = 21 h + s; k = 41: norm slot
    }
    const int64_t vtile = vt0 >> 5;
    float b0[3][21], b1[3][21];
    TileMeta m0, m1;
    f32x16 acc0[3], acc1[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) acc1[j] = f32x16{0};
    m1.ok = false; m1.ic = 0; m1.thr = 0.f;
    int64_t at = a_begin;
    load_tile_b(a, at, a_end, r, h, fetch_index(a, at, a_end, r), b0, m0);
    int32_t ic_next = fetch_index(a, at + 32, a_end, r);        // frame numbers of tile t+1
    // The prefetch is unconditional (past the end it re-reads the clamped last row, flagged !ok)
    // so that loads, epilogue and MFMAs share one basic block and can be interleaved.
    while (true) {
      // ---- even phase: MFMAs of (b0, m0) -> acc0; epilogue of (acc1, m1); prefetch into (b1, m1)
      {
        const TileMeta prev = m1;
        retire_loads(b0, m0);
        asm volatile("" ::"v"(ic_next));
        load_tile_b(a, at + 32, a_end, r, h, ic_next, b1, m1);
        ic_next = fetch_index(a, at + 64, a_end, r);
#pragma unroll
        for (int j = 0; j < 3; ++j) acc0[j] = f32x16{0};
        mfma_tile_f32(A, b0, acc0, acc1, prev, a.thr, sk, a, lane, h, vtile);
        at += 32;
        if (at >= a_end) {      // drain: epilogue of the last tile
          uint32_t m = 0;
#pragma unroll
          for (int g = 0; g < 16; ++g) m |= threshold_row(acc0, m0, g, a.thr);
          emit_tile(sk, a, lane, h, vtile, m0.ok ? m : 0u, m0.ic);
          break;
        }
      }
      // ---- odd phase: MFMAs of (b1, m1) -> acc1; epilogue of (acc0, m0); prefetch into (b0, m0)
      {
        const TileMeta prev = m0;
        retire_loads(b1, m1);
        asm volatile("" ::"v"(ic_next));
        load_tile_b(a, at + 32, a_end, r, h, ic_next, b0, m0);
        ic_next = fetch_index(a, at + 64, a_end, r);
#pragma unroll
        for (int j = 0; j < 3; ++j) acc1[j] = f32x16{0};
        mfma_tile_f32(A, b1, acc1, acc0, prev, a.thr, sk, a, lane, h, vtile);
        at += 32;
        if (at >= a_end) {
          uint32_t m = 0;
#pragma unroll
          for (int g = 0; g < 16; ++g) m |= threshold_row(acc1, m1, g, a.thr);
          emit_tile(sk, a, lane, h, vtile, m1.ok ? m : 0u, m1.ic);
          break;
        }
      }
    }
  }
  sink_flush(sk, a, lane);
}

// ------------------------------------------------------------------------------------------
// similarity GEMM, bf16 inputs / f32 accumulate on v_mfma_f32_32x32x16_bf16 (a PREFILTER: every
// survivor is re-verified in float64 by k_verify).
// K = 41 padded to 48 = three K-steps of 16, permuted so that lane half h, step s, element e holds
// k = 24 h + 8 s + e (each lane's operands are 24 consecutive bf16; two shifted copies of the row
// make any start 4-byte aligned).  Spare slots k = 42, 43 carry the audio window norm split into
// two bf16 (A holds 1 there): the accumulator ends as |A| (1 - corr) and the epilogue per video row
// is three VALU instructions: p = a0 a1;  d = p a2 - thr |A|_0 |A|_1 |A|_2;  mask = (mask << 1) | sign(d).
//
// Workgroup = 8 waves, one per CU (152 KB of LDS), two waves per SIMD (256 VGPRs each):
//   4 consumer waves, one per SIMD, 64 video rows each (two 32-row MFMA tiles, A operand resident in
//     72 VGPRs).  A consumer is alone on its SIMD's matrix pipe and software-pipelines ITSELF: the
//     threshold epilogue of the previous row tile (its own second accumulator set) and the LDS reads
//     of the next column tile's fragments are interleaved between the 9 MFMAs of the current row tile;
//   4 producer waves, one per SIMD, never touch the matrix pipe: they stage the streamed audio operand,
//     shared by all consumers, in LDS in MFMA fragment order [buffer][tile][feature][step][lane]
//     (1 KiB per fragment, conflict-free ds_read_b128) by direct global->LDS DMA (per-lane source =
//     the unaligned run), patch the two norm slots and the per-column threshold, and publish a group
//     of 8 column tiles per barrier; two LDS buffers: the DMA of group g+1 has all of group g's
//     MFMA time (~4600 cycles) to land.
// ------------------------------------------------------------------------------------------
constexpr int kBfGroup = 8;                           // 32-column tiles per staged group
constexpr int kBfTileBytes = 9 * 1024;                // 3 features x 3 steps x 64 lanes x 16 B
constexpr int kBfBufBytes = kBfGroup * kBfTileBytes;  // one group
constexpr int kBfBuffers = 2;
constexpr int kBfSurv = 256;                          // survivor staging slots per consumer wave (flushed when the next column tile could overflow it)
constexpr int kBfConsumers = 4;                       // consumer waves per workgroup
constexpr int kBfRowTiles = 2;                        // 32-row MFMA tiles per consumer
constexpr int kBfProducers = 4;                       // producer waves (two tiles of every group each)
constexpr int kBfThreads = 64 * (kBfConsumers + kBfProducers);
constexpr int kBfRowsPerBlock = 32 * kBfRowTiles * kBfConsumers;
constexpr int kBfSurvStride = kBfSurv + 64;           // + one scrap slot per lane (lanes without a survivor write there)
constexpr int kBfSmem = kBfBuffers * kBfBufBytes + kBfConsumers * kBfSurvStride * 8 + kBfBuffers * kBfGroup * 32 * (4 + 4);
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gbl_ptr_t;

__device__ __forceinline__ void stage_tile_bf16(const MatchArgs& a, unsigned char* tile_lds, int32_t ic, int h) {
  const int32_t st = ic + 24 * h;                // first element of this lane's run
  const int32_t odd = st & 1;
  const int32_t ev = st - odd;                   // even element index into the chosen copy
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    const uint16_t* base = (odd ? a.bfa_odd[j] : a.bfa_even[j]) + ev;
#pragma unroll
    for (int s = 0; s < 3; ++s) {
      unsigned char* dst = tile_lds + (3 * j + s) * 1024;        // + lane*16 added by the hardware
      __builtin_amdgcn_global_load_lds((gbl_ptr_t)(base + 8 * s), (lds_ptr_t)dst, 16, 0, 0);
    }
  }
}

#ifdef DA_DBG_STAMPS
__device__ unsigned long long g_stamps[16];
#endif

// threshold epilogue of one accumulator row of a finished tile: three VALU instructions.
// The three accumulator blocks of a tile are 16 registers apart, i.e. in the same VGPR bank for equal
// row index g, and two operands from one bank cost the instruction an extra cycle.  Feature j's A
// operand is therefore built with its rows rotated by j inside every group of four (bf_arow), so that
// video row g's three values sit in registers g, g^+1, g^+2 (rotation inside the group of four): three
// different banks.
__device__ __forceinline__ constexpr int bf_rot(int g, int j) { return (g & ~3) | ((g + j) & 3); }
// video row (0..31 within the MFMA tile) that lane r of feature j's A operand carries
__device__ __forceinline__ int bf_arow(int r, int j) { return (r & ~3) | ((r - j) & 3); }
__device__ __forceinline__ void bf_row(const f32x16 (&acc)[3], int g, float thr, uint32_t& mask) {
  const float p = acc[0][g] * acc[1][bf_rot(g, 1)];
  const float d = __builtin_fmaf(p, acc[2][bf_rot(g, 2)], -thr);    // < 0  <=>  a0 a1 a2 < thr |A|0 |A|1 |A|2
  mask = __builtin_amdgcn_alignbit(mask, __float_as_uint(d), 31);   // (mask << 1) | sign(d): rows fed 15..0
}

// Survivors of one finished tile, straight-line (no branch: it sits between two runs of MFMAs and
// is scheduled into them): every lane with a non-zero row mask stages one record at the next free
// slot of its wave's LDS buffer; the others write to their own scrap slot behind it.  The caller flushes
// the buffer before it can overflow (bf_flush_if_needed).
__device__ __forceinline__ void bf_emit(SurvSink& sk, int h, int64_t vtile, uint32_t mask, int32_t ic) {
  const unsigned long long m = __ballot(mask != 0u);
  if (m == 0ull) return;               // the wave is instruction-issue bound: ~4 in 10 tiles have no survivor at all
  const int pos = sk.count + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
  // record = ic << 41 | vtile << 17 | h << 16 | mask, as two dwords (vtile < 2^24: its bits end at 40)
  const uint32_t lo = mask | ((uint32_t)h << 16) | ((uint32_t)vtile << 17);
  const uint32_t hi = ((uint32_t)ic << 9) | (uint32_t)(vtile >> 15);
  if (mask != 0u) reinterpret_cast<uint2*>(sk.s_buf)[pos] = uint2{lo, hi};
  sk.count += __popcll(m);
}
__device__ __forceinline__ void bf_flush_if_needed(SurvSink& sk, const MatchArgs& a, int lane) {
  if (sk.count > kBfSurv - 128) sink_flush(sk, a, lane);          // a column tile (two row tiles) adds at most 128 records
}

// 9 MFMAs of one (32 video rows x 32 audio columns) tile into `acc`, with the epilogue of the
// previously finished tile (`accp`, `thr_p`) -- and, when NEXT, the LDS reads of the next column
// tile's fragments into the registers the MFMAs have just consumed -- interleaved between them.
template <bool NEXT>
__device__ __forceinline__ void bf_tile(const bf16x8 (&A)[3][3], bf16x8 (&frag)[3][3], f32x16 (&acc)[3],
                                        const f32x16 (&accp)[3], float thr_p, uint32_t& mask_p,
                                        const unsigned char* next_frags) {
#pragma unroll
  for (int j = 0; j < 3; ++j) acc[j] = f32x16{0};
  int row = 15;
#pragma unroll
  for (int m = 0; m < 9; ++m) {
#ifdef DA_BF_CHAINED
    const int j = m / 3, s = m % 3;                  // three dependent MFMAs per accumulator in a row
#else
    const int j = m % 3, s = m / 3;                  // accumulators interleaved: an MFMA never waits for the one before it
#endif
    acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[j][s], frag[j][s], acc[j], 0, 0, 0);
    if (NEXT) *reinterpret_cast<uint4*>(&frag[j][s]) = *reinterpret_cast<const uint4*>(next_frags + (3 * j + s) * 1024);
#ifndef DA_DBG_BF_NOEPI
    const int nrows = m < 7 ? 2 : 1;                 // 16 rows over 9 MFMA slots
#pragma unroll
    for (int t = 0; t < nrows; ++t) { bf_row(accp, row, thr_p, mask_p); --row; }
#endif
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);               // one MFMA
    __builtin_amdgcn_sched_group_barrier(0x002, 6, 0);               // then up to six VALU
    if (NEXT) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);     // and one LDS read
  }
#ifdef DA_DBG_BF_NOEPI
  asm volatile("" ::"v"(accp[0][0]), "v"(accp[1][7]), "v"(accp[2][15]), "v"(thr_p));
#endif
}

__global__ __launch_bounds__(kBfThreads, 2) void k_match_bf16(MatchArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* s_b = smem;                                                        // [kBfBuffers][kBfGroup][9 KiB]
  unsigned long long* s_surv = reinterpret_cast<unsigned long long*>(smem + kBfBuffers * kBfBufBytes);     // [consumers][kBfSurv]
  float* s_thr = reinterpret_cast<float*>(smem + kBfBuffers * kBfBufBytes + kBfConsumers * kBfSurvStride * 8);   // [kBfBuffers][kBfGroup][32]
  int32_t* s_ic = reinterpret_cast<int32_t*>(s_thr + kBfBuffers * kBfGroup * 32);                           // [kBfBuffers][kBfGroup][32]
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int r = lane & 31, h = lane >> 5;
  const bool producer = wave >= kBfConsumers;
  const int64_t a_begin = (int64_t)blockIdx.y * a.audio_tiles_per_block * 32;
  int64_t a_end = a_begin + (int64_t)a.audio_tiles_per_block * 32;
  if (a_end > a.n_a) a_end = a.n_a;
  if (a_begin >= a_end) return;                                           // uniform per block
  const int64_t n_tiles = (a_end - a_begin + 31) / 32;
  const int64_t n_groups = (n_tiles + kBfGroup - 1) / kBfGroup;
  auto tile_pos = [&](int64_t g, int w) { return a_begin + (g * kBfGroup + w) * 32; };

  if (producer) {
    // ------------------------------------------------------------------ producer wave
    // Stages tiles pw and pw + 4 of every group, one group ahead of the consumers.
    const int pw = wave - kBfConsumers;
    auto fetch_patch = [&](int32_t ic, uint32_t (&pv)[3]) {
      if (h) { pv[0] = a.nrmpk_a[0][ic]; pv[1] = a.nrmpk_a[1][ic]; pv[2] = a.nrmpk_a[2][ic]; }
      else { pv[0] = __float_as_uint(a.prod_a[ic]); pv[1] = 0; pv[2] = 0; }
    };
    int32_t ic_c[2], ic_n[2];                            // frame numbers: group being published / group in flight
    uint32_t pv_c[2][3], pv_n[2][3];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      ic_c[u] = fetch_index(a, tile_pos(0, pw + 4 * u), a_end, r);
      fetch_patch(ic_c[u], pv_c[u]);
      stage_tile_bf16(a, s_b + (pw + 4 * u) * kBfTileBytes, ic_c[u], h);
      ic_n[u] = fetch_index(a, tile_pos(1, pw + 4 * u), a_end, r);
    }
    for (int64_t g = 0; g < n_groups; ++g) {
      const int cur = (int)(g & 1);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // the DMA of group g has landed
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int w = pw + 4 * u;
        unsigned char* tile = s_b + cur * kBfBufBytes + w * kBfTileBytes;
        if (h) {
#pragma unroll
          for (int j = 0; j < 3; ++j)
            *reinterpret_cast<uint32_t*>(tile + (3 * j + 2) * 1024 + lane * 16 + 4) = pv_c[u][j];
        } else {
          s_thr[(cur * kBfGroup + w) * 32 + r] = ((tile_pos(g, w) + r) < a_end) ? a.thr * __uint_as_float(pv_c[u][0]) : -__builtin_inff();
          s_ic[(cur * kBfGroup + w) * 32 + r] = ic_c[u];
        }
      }
      __syncthreads();                           // group g published; the consumers have left buffer cur ^ 1
      if (g + 1 < n_groups) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          fetch_patch(ic_n[u], pv_n[u]);
          stage_tile_bf16(a, s_b + (cur ^ 1) * kBfBufBytes + (pw + 4 * u) * kBfTileBytes, ic_n[u], h);
        }
      }
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        ic_c[u] = ic_n[u]; pv_c[u][0] = pv_n[u][0]; pv_c[u][1] = pv_n[u][1]; pv_c[u][2] = pv_n[u][2];
        ic_n[u] = fetch_index(a, tile_pos(g + 2, pw + 4 * u), a_end, r);
      }
    }
    return;
  }

  // -------------------------------------------------------------------- consumer wave
  const int64_t vt0 = ((int64_t)blockIdx.x * kBfConsumers + wave) * (32 * kBfRowTiles);   // this wave's 64 video rows
  SurvSink sk{s_surv + wave * kBfSurvStride, 0, kBfSurv};
  bf16x8 A[kBfRowTiles][3][3];
#pragma unroll
  for (int rt = 0; rt < kBfRowTiles; ++rt) {
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const int64_t vr = vt0 + 32 * rt + bf_arow(r, j);          // rows rotated per feature: see bf_row
      const bool vok = vr < a.n_v;
      const int32_t v = a.vlist[vok ? vr : a.n_v - 1];
      const double sc = vok ? -(double)a.inv_v[j][v] : 0.0;
      const double* p = a.msd_v[j] + v;
#pragma unroll
      for (int s = 0; s < 3; ++s)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const int k = 24 * h + 8 * s + e;
          uint16_t x = 0;
          if (k < kWin) x = f32_to_bf16((float)(p[k] * sc));
          else if (k == 42 || k == 43) x = 0x3F80;            // 1.0: the norm slots
          A[rt][j][s][e] = (short)x;
        }
    }
  }
  const int64_t vtile0 = vt0 >> 5;
  // Software pipeline over "phases" (one row tile x one column tile = 9 MFMAs).  Phase order:
  //   rt0(0) | rt1(0) rt0(1) | rt1(1) rt0(2) | ...      ( | = loop back-edge )
  // Each phase's MFMAs carry, interleaved: the threshold epilogue of the PREVIOUS phase's accumulators
  // (two accumulator sets), and -- in rt1 phases -- the LDS reads of the next column tile's fragments
  // into the registers the MFMAs have just consumed; the survivor records of the phase before that
  // are written at the start of the phase.  The loop is rotated so that its back-edge sits behind an
  // rt0 phase: every LDS operation in flight there is at least nine MFMAs old, and the
  // s_waitcnt lgkmcnt(0) the compiler places at a loop header costs nothing.
  f32x16 accX[3], accY[3];
#pragma unroll
  for (int j = 0; j < 3; ++j) accY[j] = f32x16{0};
  float thr_y = -__builtin_inff();                     // threshold row for accY's columns; -inf: nothing owed, no row passes
  int32_t ic_y = 0;
  uint32_t pend_mask = 0; int32_t pend_ic = 0; int pend_rt = 0;      // finished epilogue whose records are not staged yet
  bf16x8 frag[3][3];
#ifdef DA_DBG_STAMPS
  unsigned long long st_acc[4] = {0, 0, 0, 0};
  unsigned long long st_t = __builtin_amdgcn_s_memtime();
#define BF_STAMP(k) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); st_acc[k] += t_ - st_t; st_t = t_; }
#else
#define BF_STAMP(k)
#endif
  for (int64_t g = 0; g < n_groups; ++g) {
    const int cur = (int)(g & 1);
    BF_STAMP(0)                                                    // phases of the previous group
    __syncthreads();                                               // group g is in buffer cur
    BF_STAMP(1)                                                    // barrier wait
    const unsigned char* gbase = s_b + cur * kBfBufBytes + lane * 16;
#pragma unroll
    for (int m = 0; m < 9; ++m) *reinterpret_cast<uint4*>(&frag[m / 3][m % 3]) = *reinterpret_cast<const uint4*>(gbase + m * 1024);
#ifdef DA_DBG_STAMPS
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    BF_STAMP(2)                                                    // exposed fragment load
#endif
    int64_t left = n_tiles - g * kBfGroup;                         // column tiles of this group
    if (left > kBfGroup) left = kBfGroup;
    float thr_x = s_thr[(cur * kBfGroup) * 32 + r];                // -inf: column past the end
    int32_t ic_x = s_ic[(cur * kBfGroup) * 32 + r];
    auto stage_pending = [&]() {
#ifndef DA_DBG_BF_NOEMIT
      bf_emit(sk, h, vtile0 + pend_rt, pend_mask, pend_ic);
#else
      asm volatile("" ::"v"(pend_mask));
#endif
    };
    // rt0 of the group's first column tile; owed: rt1 of the previous group's last column tile
    {
      stage_pending();
      uint32_t mask = 0;
      bf_tile<false>(A[0], frag, accX, accY, thr_y, mask, nullptr);
      pend_mask = mask; pend_ic = ic_y; pend_rt = 1;
    }
#pragma unroll 1
    for (int w = 0; w + 1 < (int)left; ++w) {
      // rt1(w): epilogue of rt0(w); fragments of column tile w + 1 stream in
      stage_pending();
      uint32_t mask = 0;
      bf_tile<true>(A[1], frag, accY, accX, thr_x, mask, gbase + (w + 1) * kBfTileBytes);
      pend_mask = mask; pend_ic = ic_x; pend_rt = 0;
      thr_y = thr_x; ic_y = ic_x;
      thr_x = s_thr[(cur * kBfGroup + w + 1) * 32 + r];
      ic_x = s_ic[(cur * kBfGroup + w + 1) * 32 + r];
      // rt0(w + 1): epilogue of rt1(w)
      stage_pending();
      mask = 0;
      bf_tile<false>(A[0], frag, accX, accY, thr_y, mask, nullptr);
      pend_mask = mask; pend_ic = ic_y; pend_rt = 1;
#ifndef DA_DBG_BF_NOEMIT
      bf_flush_if_needed(sk, a, lane);
#endif
    }
    // rt1 of the group's last column tile (the next fragments come from the other buffer, after the barrier)
    {
      stage_pending();
      uint32_t mask = 0;
      bf_tile<false>(A[1], frag, accY, accX, thr_x, mask, nullptr);
      pend_mask = mask; pend_ic = ic_x; pend_rt = 0;
      thr_y = thr_x; ic_y = ic_x;
#ifndef DA_DBG_BF_NOEMIT
      bf_flush_if_needed(sk, a, lane);
#endif
    }
  }
#ifdef DA_DBG_STAMPS
  BF_STAMP(0)
  if (wave == 0 && lane == 0) {
    for (int t = 0; t < 3; ++t) atomicAdd(&g_stamps[t], st_acc[t]);
    atomicAdd(&g_stamps[3], (unsigned long long)(n_tiles * 2));    // phases executed
  }
#endif
  {                                                                // drain: staged record of rt0, epilogue + record of the last rt1
    bf_emit(sk, h, vtile0 + pend_rt, pend_mask, pend_ic);
    uint32_t mask = 0;
#pragma unroll
    for (int g = 15; g >= 0; --g) bf_row(accY, g, thr_y, mask);
    bf_emit(sk, h, vtile0 + 1, mask, ic_y);
  }
  sink_flush(sk, a, lane);
}

void debug_read_stamps(unsigned long long out[16]) {
  for (int k = 0; k < 16; ++k) out[k] = 0;
#ifdef DA_DBG_STAMPS
  (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_stamps), sizeof(unsigned long long) * 16);
  unsigned long long z[16] = {0};
  (void)hipMemcpyToSymbol(HIP_SYMBOL(g_stamps), z, sizeof z);
#endif
}

static dim3 match_grid(const MatchArgs& a) {
  const int64_t vtiles = (a.n_v + 31) / 32;
  const int64_t bx = (vtiles + kWavesPerBlock - 1) / kWavesPerBlock;
  const int64_t atiles = (a.n_a + 31) / 32;
  const int64_t by = (atiles + a.audio_tiles_per_block - 1) / a.audio_tiles_per_block;
  return dim3((unsigned)bx, (unsigned)by);
}

void launch_match_f32(const MatchArgs& a, hipStream_t s) {
  if (a.n_v <= 0 || a.n_a <= 0) return;
  hipLaunchKernelGGL(k_match_f32, match_grid(a), dim3(64 * kWavesPerBlock), 0, s, a);
}
void launch_match_bf16(const MatchArgs& a, hipStream_t s) {
  if (a.n_v <= 0 || a.n_a <= 0) return;
  // per launch: the attribute is per device and contexts on several devices / threads share this code
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_match_bf16), hipFuncAttributeMaxDynamicSharedMemorySize, kBfSmem);
  const int64_t bx = (a.n_v + kBfRowsPerBlock - 1) / kBfRowsPerBlock;
  const int64_t atiles = (a.n_a + 31) / 32;
  const int64_t by = (atiles + a.audio_tiles_per_block - 1) / a.audio_tiles_per_block;
  hipLaunchKernelGGL(k_match_bf16, dim3((unsigned)bx, (unsigned)by), dim3(kBfThreads), kBfSmem, s, a);
}

// diagnostics: the raw MFMA accumulators of ONE (32 video rows x 32 audio columns) tile, formed with
// the production kernels' own operand construction and instruction sequence (K permutation, norm
// slots, bf16 rounding, per-feature row rotation): out[j][row][col] = |A|_j(col) (1 - corr_j(row, col)).
// One wavefront.  The matrix instructions are deterministic, so these are the values the threshold
// epilogue of k_match_f32 / k_match_bf16 sees for that tile.
__global__ __launch_bounds__(64) void k_dump_tile(MatchArgs a, int64_t vtile, int64_t atile, int bf16, float* __restrict__ out,
                                                  int32_t* __restrict__ vframes, int32_t* __restrict__ aframes) {
  const int lane = threadIdx.x & 63;
  const int r = lane & 31, h = lane >> 5;
  const int64_t at = atile * 32;
  const int32_t ic = fetch_index(a, at, a.n_a, r);
  f32x16 acc[3];
#pragma unroll
  for (int j = 0; j < 3; ++j) acc[j] = f32x16{0};
  if (!bf16) {
    const int64_t vr = vtile * 32 + r;
    const bool vok = vr < a.n_v;
    const int32_t v = a.vlist[vok ? vr : a.n_v - 1];
    float A[3][21], b[3][21];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const float sc = vok ? -a.inv_v[j][v] : 0.f;
      const float* p = a.ms_v[j] + v + 21 * h;
#pragma unroll
      for (int s = 0; s < 21; ++s) A[j][s] = (21 * h + s < kWin) ? p[s] * sc : 1.0f;
    }
    TileMeta m;
    load_tile_b(a, at, a.n_a, r, h, ic, b, m);
#pragma unroll
    for (int s = 0; s < 21; ++s)
#pragma unroll
      for (int j = 0; j < 3; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(A[j][s], b[j][s], acc[j], 0, 0, 0);
    if (h == 0) { vframes[r] = vok ? v : -1; aframes[r] = (at + r) < a.n_a ? ic : -1; }
  } else {
    bf16x8 A[3][3], B[3][3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const int64_t vr = vtile * 32 + bf_arow(r, j);
      const bool vok = vr < a.n_v;
      const int32_t v = a.vlist[vok ? vr : a.n_v - 1];
      const double sc = vok ? -(double)a.inv_v[j][v] : 0.0;
      const double* p = a.msd_v[j] + v;
      const uint32_t pk = a.nrmpk_a[j][ic];
#pragma unroll
      for (int s = 0; s < 3; ++s)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const int k = 24 * h + 8 * s + e;
          uint16_t x = 0;
          if (k < kWin) x = f32_to_bf16((float)(p[k] * sc));
          else if (k == 42 || k == 43) x = 0x3F80;
          A[j][s][e] = (short)x;
          uint16_t y = a.bfa_even[j][ic + k];                         // what the LDS-DMA stages (either copy holds the same values)
          if (k == 42) y = (uint16_t)(pk & 0xFFFFu);                  // the producers' norm patch: hi, lo halves of |A|
          if (k == 43) y = (uint16_t)(pk >> 16);
          B[j][s][e] = (short)y;
        }
    }
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
      for (int s = 0; s < 3; ++s) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[j][s], B[j][s], acc[j], 0, 0, 0);
    if (h == 0) {
      const int64_t vr = vtile * 32 + r;
      vframes[r] = vr < a.n_v ? a.vlist[vr] : -1;
      aframes[r] = (at + r) < a.n_a ? ic : -1;
    }
  }
#pragma unroll
  for (int g = 0; g < 16; ++g) {
    const int row = (g & 3) + 8 * (g >> 2) + 4 * h;
#pragma unroll
    for (int j = 0; j < 3; ++j)
      out[(j * 32 + row) * 32 + r] = bf16 ? acc[j][bf_rot(g, j)] : acc[j][g];
  }
}
void launch_dump_tile(const MatchArgs& a, int64_t vtile, int64_t atile, int bf16, float* d_out, int32_t* d_vframes, int32_t* d_aframes,
                      hipStream_t s) {
  hipLaunchKernelGGL(k_dump_tile, dim3(1), dim3(64), 0, s, a, vtile, atile, bf16, d_out, d_vframes, d_aframes);
}

// diagnostics: the correlations exactly as the GEMM precision forms them, for explicit pairs.
// One wave per 32 pairs would be the MFMA way; this path is test-only, so it uses plain FMAs on
// identically rounded operands (f32: same k-ordered fmaf chain as the MFMA; bf16: same rounding
// of both operands, f32 accumulation).
__global__ void k_corr(CorrArgs c) {
  const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= c.n) return;
  const int32_t i = c.pi[p], v = c.pv[p];
  for (int j = 0; j < 3; ++j) {
    float acc = 0.f;
    if (c.precision == 0) {
      const float sc = -c.m.inv_v[j][v];
      for (int k = 0; k < kWin; ++k) acc = fmaf(c.m.ms_v[j][v + k] * sc, c.m.ms_a[j][i + k], acc);
    } else {
      const double sc = -(double)c.m.inv_v[j][v];
      for (int k = 0; k < kWin; ++k) {
        const uint16_t ab = f32_to_bf16((float)(c.m.msd_v[j][v + k] * sc));
        const uint16_t bb = c.m.bfa_even[j][i + k];
        const float af = __uint_as_float((uint32_t)ab << 16), bf = __uint_as_float((uint32_t)bb << 16);
        acc = fmaf(af, bf, acc);
      }
    }
    c.corr[3 * p + j] = -acc * c.m.inv_a[j][i];
  }
}
void launch_corr(const CorrArgs& a, hipStream_t s) {
  if (a.n <= 0) return;
  hipLaunchKernelGGL(k_corr, dim3((unsigned)((a.n + 255) / 256)), dim3(256), 0, s, a);
}

// ------------------------------------------------------------------------------------------
// exact verification of survivors (float64), hash vote, quality (:649-673)
// ------------------------------------------------------------------------------------------
// exact float64 re-evaluation of one pair; returns true and the quality when it is a match
__device__ inline bool verify_pair(const VerifyArgs& a, int32_t i, int32_t v, double& q_out) {
  if (a.mode == 0) {
    // (feature 3 or 4 hits) and (at least two of features 0-2 hit), cheapest rejection first: a
    // hash hit of one feature is a ~1e-3 event for an unrelated pair, so testing the two-feature
    // alternative first rejects almost every survivor after 3-6 gathers instead of 9-15
    const bool h3 = digit_hit(a.dig_a[3][i], a.dig_v[3][v], a.flg_v[3][v]);
    const bool h34 = h3 ? true : digit_hit(a.dig_a[4][i], a.dig_v[4][v], a.flg_v[4][v]);
    if (!h34) return false;
    const int h0 = digit_hit(a.dig_a[0][i], a.dig_v[0][v], a.flg_v[0][v]) ? 1 : 0;
    const int h1 = digit_hit(a.dig_a[1][i], a.dig_v[1][v], a.flg_v[1][v]) ? 1 : 0;
    if (h0 + h1 == 0) return false;
    if (h0 + h1 < 2 && !digit_hit(a.dig_a[2][i], a.dig_v[2][v], a.flg_v[2][v])) return false;
  }
  double prob = 1.0;
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    const double* pa = a.ms_a[j] + i;
    const double* pv = a.ms_v[j] + v;
    double dot = 0.0;
#pragma unroll
    for (int k = 0; k < kWin; ++k) dot = fma(pa[k], pv[k], dot);
    const double corr = dot / (a.nrm_a[j][i] * a.nrm_v[j][v]);
    const double t = 1.0 - corr;
    prob *= (t > 1e-8 ? t : 1e-8);
  }
  prob = pow(prob, 2.9);
  if (prob > 1e-8) return false;
  const double q = pow(prob / 1e-12, -1.0 / 3.0);
  q_out = q < 50.0 ? q : 50.0;
  return true;
}

// Grid-stride over the staged records: expand each row mask into (i, v) pairs, verify, and stage
// the matches of the whole block in LDS; output space is reserved with ONE global atomic per
// flush (a single hot counter sustains only ~90 atomics/us, and there are up to 1e9 pairs).
constexpr int kVerifyThreads = 256;
constexpr int kVerifyStage = 1024;

__global__ __launch_bounds__(kVerifyThreads) void k_verify(VerifyArgs a, unsigned long long n_rec) {
  __shared__ unsigned long long s_key[kVerifyStage];
  __shared__ double s_q[kVerifyStage];
  __shared__ unsigned int s_n;
  __shared__ unsigned long long s_base;
  if (threadIdx.x == 0) s_n = 0;
  __syncthreads();
  const unsigned long long stride = (unsigned long long)gridDim.x * kVerifyThreads;
  const unsigned long long rounds = (n_rec + stride - 1) / stride;
  for (unsigned long long rnd = 0; rnd < rounds; ++rnd) {
    const unsigned long long p = rnd * stride + (unsigned long long)blockIdx.x * kVerifyThreads + threadIdx.x;
    const unsigned long long rec = (p < n_rec) ? a.surv[p] : 0ull;
    const int32_t i = (int32_t)(rec >> 41);
    const int64_t vtile = (int64_t)((rec >> 17) & 0xFFFFFFull);
    const int h = (int)((rec >> 16) & 1ull);
    uint32_t mask = (uint32_t)(rec & 0xFFFFull);
    while (mask != 0u) {
      const int g = __ffs(mask) - 1;
      mask &= mask - 1u;
      const int row = (g & 3) + 8 * (g >> 2) + 4 * h;
      const int64_t vr = vtile * 32 + row;
      double q;
      if (vr < a.n_v) {
        const int32_t v = a.vlist[vr];
        if (verify_pair(a, i, v, q)) {
          const unsigned int pos = atomicAdd(&s_n, 1u);
          if (pos < (unsigned)kVerifyStage) {
            s_key[pos] = ((unsigned long long)(uint32_t)i << 32) | (uint32_t)v;
            s_q[pos] = q;
          } else {                                   // stage full: rare, go straight to global
            const unsigned long long gp = atomicAdd(a.n_out, 1ull);
            if (gp < a.out_capacity) { a.keys[gp] = ((unsigned long long)(uint32_t)i << 32) | (uint32_t)v; a.quals[gp] = q; }
          }
        }
      }
    }
    __syncthreads();
    // flush when the stage could overflow in the next round (each thread adds at most 16)
    const unsigned int n = s_n < (unsigned)kVerifyStage ? s_n : (unsigned)kVerifyStage;
    const bool last = (rnd + 1 == rounds);
    if (n > (unsigned)(kVerifyStage / 2) || (last && n > 0)) {
      if (threadIdx.x == 0) s_base = atomicAdd(a.n_out, (unsigned long long)n);
      __syncthreads();
      for (unsigned int t = threadIdx.x; t < n; t += kVerifyThreads) {
        const unsigned long long gp = s_base + t;
        if (gp < a.out_capacity) { a.keys[gp] = s_key[t]; a.quals[gp] = s_q[t]; }
      }
      __syncthreads();
      if (threadIdx.x == 0) s_n = 0;
      __syncthreads();
    }
  }
}

__global__ void k_unpack_keys(const unsigned long long* __restrict__ keys, int64_t n, int32_t* __restrict__ oi, int32_t* __restrict__ ov) {
  const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n) return;
  const unsigned long long k = keys[p];
  oi[p] = (int32_t)(k >> 32);
  ov[p] = (int32_t)(k & 0xffffffffu);
}
void launch_unpack_keys(const unsigned long long* keys, int64_t n, int32_t* out_i, int32_t* out_v, hipStream_t s) {
  if (n <= 0) return;
  hipLaunchKernelGGL(k_unpack_keys, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, keys, n, out_i, out_v);
}

void launch_verify(const VerifyArgs& a, unsigned long long n_surv_host, hipStream_t s) {
  if (n_surv_host == 0) return;
  unsigned long long blocks = (n_surv_host + kVerifyThreads - 1) / kVerifyThreads;
  if (blocks > 256 * 16) blocks = 256 * 16;
  hipLaunchKernelGGL(k_verify, dim3((unsigned)blocks), dim3(kVerifyThreads), 0, s, a, n_surv_host);
}

// ------------------------------------------------------------------------------------------
// pass 2: evaluation along a cluster's line (:901-906, :916-936), float64
// ------------------------------------------------------------------------------------------
__device__ inline void interp3(const double* __restrict__ vs, int64_t Lv, double y, double out[3]) {
  // k=1 spline on integer knots (:864): linear interpolation between neighbouring frames
  double fl = floor(y);
  int64_t k = (int64_t)fl;
  if (k < 0) { k = 0; }
  if (k > Lv - 2) { k = Lv - 2; }
  const double t = y - (double)k;
  const double* p = vs + 3 * k;
#pragma unroll
  for (int c = 0; c < 3; ++c) out[c] = p[c] * (1.0 - t) + p[3 + c] * t;
}

__global__ __launch_bounds__(256) void k_band_refine(BandArgs a, double* __restrict__ partials) {
  // rows x in (lo, hi-1) exclusive of both ends (the reference drops the first and last, :918)
  double cnt = 0, sde = 0, sdd = 0, see = 0;
  for (int64_t x = a.lo + 1 + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; x < a.hi - 1;
       x += (int64_t)gridDim.x * blockDim.x) {
    double vm[3], vp[3], vn[3];
    interp3(a.v_scaled, a.Lv, a.slope * (double)x + a.offset, vm);
    interp3(a.v_scaled, a.Lv, a.slope * (double)(x - 1) + a.offset, vp);
    interp3(a.v_scaled, a.Lv, a.slope * (double)(x + 1) + a.offset, vn);
    const double* am = a.a_scaled + 3 * x;
    double e[3], mean = 0.0;
#pragma unroll
    for (int c = 0; c < 3; ++c) { e[c] = am[c] - vm[c]; mean += e[c]; }
    mean /= 3.0;
    if (mean < 0.1) {
      cnt += 1.0;
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const double d = (vn[c] - vp[c]) / 2.0;
        sde = fma(d, e[c], sde); sdd = fma(d, d, sdd); see = fma(e[c], e[c], see);
      }
    }
  }
  __shared__ double red[4][256];
  red[0][threadIdx.x] = cnt; red[1][threadIdx.x] = sde; red[2][threadIdx.x] = sdd; red[3][threadIdx.x] = see;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s)
      for (int c = 0; c < 4; ++c) red[c][threadIdx.x] += red[c][threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0)
    for (int c = 0; c < 4; ++c) partials[4 * blockIdx.x + c] = red[c][0];
}
void launch_band_refine(const BandArgs& a, double* d_partials, int n_blocks, hipStream_t s) {
  hipLaunchKernelGGL(k_band_refine, dim3(n_blocks), dim3(256), 0, s, a, d_partials);
}

__global__ __launch_bounds__(256) void k_band_quality(BandArgs a, double* __restrict__ ys, double* __restrict__ qs) {
  const int64_t x = a.lo + (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (x >= a.hi) return;
  const double y = a.slope * (double)x + a.offset;
  double vm[3];
  interp3(a.v_scaled, a.Lv, y, vm);
  const double* am = a.a_scaled + 3 * x;
  double q = 0.0;
#pragma unroll
  for (int c = 0; c < 3; ++c) q += -0.5 - log10(1e-4 + fabs(am[c] - vm[c]));
  double g = vm[0] + 2.5 - a.v_max; g = g < 0 ? 0 : (g > 1 ? 1 : g);
  q *= g;
  double ga = am[0] + 2.5 - a.a_max; ga = ga < 0 ? 0 : (ga > 1 ? 1 : ga);
  q += ga * 0.1;
  ys[x - a.lo] = y;
  qs[x - a.lo] = q;
}
void launch_band_quality(const BandArgs& a, double* d_y, double* d_q, hipStream_t s) {
  const int64_t n = a.hi - a.lo;
  if (n <= 0) return;
  hipLaunchKernelGGL(k_band_quality, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, a, d_y, d_q);
}

// ---- pass 2, all clusters in one launch -------------------------------------------------------
// blockIdx.y = cluster: the sub-frame refinement sums of every long cluster (:916-930)
__global__ __launch_bounds__(256) void k_band_refine_all(BandArgs base, const BandCluster* __restrict__ cl, double* __restrict__ partials) {
  const BandCluster c = cl[blockIdx.y];
  double cnt = 0, sde = 0, sdd = 0, see = 0;
  if (c.refine) {
    BandArgs a = base;
    a.offset = c.offset; a.slope = c.slope; a.lo = c.lo; a.hi = c.hi;
    for (int64_t x = a.lo + 1 + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; x < a.hi - 1;
         x += (int64_t)gridDim.x * blockDim.x) {
      double vm[3], vp[3], vn[3];
      interp3(a.v_scaled, a.Lv, a.slope * (double)x + a.offset, vm);
      interp3(a.v_scaled, a.Lv, a.slope * (double)(x - 1) + a.offset, vp);
      interp3(a.v_scaled, a.Lv, a.slope * (double)(x + 1) + a.offset, vn);
      const double* am = a.a_scaled + 3 * x;
      double e[3], mean = 0.0;
#pragma unroll
      for (int k = 0; k < 3; ++k) { e[k] = am[k] - vm[k]; mean += e[k]; }
      mean /= 3.0;
      if (mean < 0.1) {
        cnt += 1.0;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          const double d = (vn[k] - vp[k]) / 2.0;
          sde = fma(d, e[k], sde); sdd = fma(d, d, sdd); see = fma(e[k], e[k], see);
        }
      }
    }
  }
  __shared__ double red[4][256];
  red[0][threadIdx.x] = cnt; red[1][threadIdx.x] = sde; red[2][threadIdx.x] = sdd; red[3][threadIdx.x] = see;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s)
      for (int k = 0; k < 4; ++k) red[k][threadIdx.x] += red[k][threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0)
    for (int k = 0; k < 4; ++k) partials[4 * ((size_t)blockIdx.y * gridDim.x + blockIdx.x) + k] = red[k][0];
}
void launch_band_refine_all(const BandArgs& base, const BandCluster* d_cl, int n_clusters, double* d_partials, int n_blocks, hipStream_t s) {
  if (n_clusters <= 0) return;
  hipLaunchKernelGGL(k_band_refine_all, dim3(n_blocks, n_clusters), dim3(256), 0, s, base, d_cl, d_partials);
}

// one thread per banded point of ANY cluster (clusters laid out one after another: first[c] = prefix of
// their lengths): video position y, quality (:931-936), and the key (audio frame << 32 | int(y)) the
// reference de-duplicates on (:937-941)
__global__ __launch_bounds__(256) void k_band_quality_all(BandArgs base, const BandCluster* __restrict__ cl, int n_clusters, int64_t n_points,
                                                          double* __restrict__ ys, double* __restrict__ qs, int32_t* __restrict__ cls,
                                                          unsigned long long* __restrict__ keys, int32_t* __restrict__ ids) {
  const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n_points) return;
  int lo = 0, hi = n_clusters - 1;                      // last cluster with first <= p
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (cl[mid].first <= p) lo = mid; else hi = mid - 1;
  }
  const BandCluster c = cl[lo];
  const int64_t x = c.lo + (p - c.first);
  const double y = c.slope * (double)x + c.offset;
  double vm[3];
  interp3(base.v_scaled, base.Lv, y, vm);
  const double* am = base.a_scaled + 3 * x;
  double q = 0.0;
#pragma unroll
  for (int k = 0; k < 3; ++k) q += -0.5 - log10(1e-4 + fabs(am[k] - vm[k]));
  double g = vm[0] + 2.5 - base.v_max; g = g < 0 ? 0 : (g > 1 ? 1 : g);
  q *= g;
  double ga = am[0] + 2.5 - base.a_max; ga = ga < 0 ? 0 : (ga > 1 ? 1 : ga);
  q += ga * 0.1;
  ys[p] = y; qs[p] = q; cls[p] = lo;
  keys[p] = ((unsigned long long)(uint32_t)x << 32) | (uint32_t)(int64_t)y;
  ids[p] = (int32_t)p;
}
void launch_band_quality_all(const BandArgs& base, const BandCluster* d_cl, int n_clusters, int64_t n_points, double* d_y, double* d_q,
                             int32_t* d_cl_of, unsigned long long* d_keys, int32_t* d_ids, hipStream_t s) {
  if (n_points <= 0) return;
  hipLaunchKernelGGL(k_band_quality_all, dim3((unsigned)((n_points + 255) / 256)), dim3(256), 0, s, base, d_cl, n_clusters, n_points,
                     d_y, d_q, d_cl_of, d_keys, d_ids);
}

// after the stable sort by key: the first point of every key is the one the reference keeps (clusters
// are visited in order); gather the kept points' fields in key order = (audio frame, video position) order
__global__ __launch_bounds__(256) void k_band_heads(const unsigned long long* __restrict__ keys, int64_t n, uint8_t* __restrict__ head) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t < n) head[t] = (t == 0 || keys[t] != keys[t - 1]) ? 1 : 0;
}
__global__ __launch_bounds__(256) void k_band_gather(const int32_t* __restrict__ kept, const int32_t* __restrict__ n_kept,
                                                     const double* __restrict__ ys, const double* __restrict__ qs, const int32_t* __restrict__ cls,
                                                     const unsigned long long* __restrict__ keys_unsorted, double* __restrict__ o_j,
                                                     double* __restrict__ o_q, int32_t* __restrict__ o_i, int32_t* __restrict__ o_cl) {
  const int32_t n = *n_kept;
  for (int32_t t = blockIdx.x * blockDim.x + threadIdx.x; t < n; t += gridDim.x * blockDim.x) {
    const int32_t p = kept[t];
    o_j[t] = ys[p]; o_q[t] = qs[p]; o_cl[t] = cls[p]; o_i[t] = (int32_t)(keys_unsorted[p] >> 32);
  }
}
void launch_band_heads(const unsigned long long* d_sorted_keys, int64_t n, uint8_t* d_head, hipStream_t s) {
  if (n <= 0) return;
  hipLaunchKernelGGL(k_band_heads, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, d_sorted_keys, n, d_head);
}
void launch_band_gather(const int32_t* d_kept, const int32_t* d_n_kept, const double* d_y, const double* d_q, const int32_t* d_cl_of,
                        const unsigned long long* d_keys_unsorted, double* o_j, double* o_q, int32_t* o_i, int32_t* o_cl, hipStream_t s) {
  hipLaunchKernelGGL(k_band_gather, dim3(512), dim3(256), 0, s, d_kept, d_n_kept, d_y, d_q, d_cl_of, d_keys_unsorted, o_j, o_q, o_i, o_cl);
}

__global__ void k_colmax(const double* __restrict__ d, int64_t n, int stride, double* out) {
  __shared__ double red[256];
  double m = -1e300;
  for (int64_t i = threadIdx.x; i < n; i += 256) m = fmax(m, d[i * stride]);
  red[threadIdx.x] = m;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) red[threadIdx.x] = fmax(red[threadIdx.x], red[threadIdx.x + s]);
    __syncthreads();
  }
  if (threadIdx.x == 0) *out = red[0];
}
void launch_colmax(const double* d, int64_t n, int stride, double* d_out, hipStream_t s) {
  hipLaunchKernelGGL(k_colmax, dim3(1), dim3(256), 0, s, d, n, stride, d_out);
}

}  // namespace da
