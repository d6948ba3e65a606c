// Audio replacement path (--stretch_audio) for gfx950: describealign.py:230-416, :1135-1153.
//
//   resampling   k_resample_tile one workgroup per tile of 4064 spline coefficients: float16 samples
//                                to LDS, Thomas recurrences with 20-row warm-ups (exact boundary
//                                rows at the true ends), coefficients stay in LDS, the same workgroup
//                                evaluates the output points on its rows (scipy make_interp_spline k=2)
//                k_spline_solve / k_spline_eval   sequential fallback for splines shorter than 64 rows
//   stretching   k_chunk_rms     wavefront per chunk of 50 windows: sliding 512-sums of the power in
//                                the reference's summation order, epsilon, rms          (:272-279)
//                k_lag_table     wavefront per (chunk, lag): lag products, sliding 512-sums (lane 0
//                                extends the float64 running sum in order, all lanes share the
//                                per-position work), Pearson correlation, arg-max per window
//                                                                                   (:280-296, :321-322)
//                k_viterbi       one workgroup per stretched interval: (window, drift) Viterbi with
//                                the 3 x 3073 cost history (+inf guard zones) in LDS, back-pointers
//                                in HBM, then the back-track and the copy plan      (:311-368)
//                k_splice        one thread per output sample: run copy + Hann cross-fades (:369-385)
//   bracket      k_pcm_moments / k_pcm_to_f16 (:156, :1135-1148), k_absmax / k_finish (:1153, :136)
//
// All of it is HBM/latency-bound byte and fp64 work; nothing here is GEMM-shaped.
#include "dalign_stretch.h"
#include "dalign_common.h"

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>

#include "../../include/dalign.h"

namespace da {

namespace {

constexpr int kSW = 512;                  // window_size                     (:251, :298)
constexpr int kMaxDrift = 3 * kSW;        // max_drift                       (:298)
constexpr int kND = 2 * kMaxDrift + 1;    // drift_window_size = 3073        (:299)
constexpr int kCached = 50;               // max_cached_chunks               (:254)
constexpr int kResChunk = 100000;         // chunk_size                      (:234)
constexpr int kMinOffset = 30;            // MIN_STRETCH_OFFSET              (:36)
constexpr int kRate = 44100;              // AUDIO_SAMPLE_RATE               (:31)
constexpr int kMaxLags = 512;
constexpr int kGuardLo = 512;            // +inf entries below drift state 0 (a lag is < 512)
constexpr int kGuardHi = 128;            // +inf entries above drift state 3072 (two drift steps)
constexpr int kRow = kGuardLo + kND + kGuardHi + 3;   // 3716: one cost-history row in LDS

typedef _Float16 half_t;

// float64 -> float16 with a single rounding (numpy's astype does the same): round to odd in
// float32 first, then the hardware's round-to-nearest-even float32 -> float16.
__device__ __forceinline__ half_t to_half(double x) {
  float f = (float)x;
  const double back = (double)f;
  if (back != x && fabsf(f) != INFINITY) {
    uint32_t u = __float_as_uint(f);
    if (fabs(back) > fabs(x)) u -= 1u;
    u |= 1u;
    f = __uint_as_float(u);
  }
  return (half_t)f;
}

struct DBuf {
  void* p = nullptr; size_t cap = 0;
  hipError_t ensure(size_t bytes) {
    if (bytes <= cap) return hipSuccess;
    if (p) { (void)hipFree(p); p = nullptr; cap = 0; }
    const size_t want = bytes + bytes / 8 + 4096;
    hipError_t e = hipMalloc(&p, want);
    if (e != hipSuccess) { p = nullptr; return e; }
    cap = want; return hipSuccess;
  }
  void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
  template <class T> T* as() const { return reinterpret_cast<T*>(p); }
};

// ------------------------------------------------------------------------------------ bracket
// The loudness / peak steps around replace_aligned_segments are pure streaming passes.  Every
// thread handles 8 consecutive frames with 16-byte accesses (scalar only where a pointer is not
// 16-byte aligned or at the ragged end); float16 device arrays are planar with a channel stride
// that is a multiple of 64 elements so that every channel starts aligned.

struct Pcm { const int16_t* p; int64_t n; int planar; };

template <int C>
__device__ __forceinline__ void load_frames8(const Pcm& s, int64_t i0, int16_t (&v)[C][8]) {
  const int64_t left = s.n - i0;
  if (C == 2 && !s.planar) {
    const int16_t* p = s.p + 2 * i0;
    if (left >= 8 && (reinterpret_cast<uintptr_t>(p) & 15) == 0) {
      const int4 a = reinterpret_cast<const int4*>(p)[0], b = reinterpret_cast<const int4*>(p)[1];
      const int w[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
#pragma unroll
      for (int k = 0; k < 8; ++k) { v[0][k] = (int16_t)(w[k] & 0xffff); v[C - 1][k] = (int16_t)(w[k] >> 16); }
    } else {
#pragma unroll
      for (int k = 0; k < 8; ++k) { v[0][k] = k < left ? p[2 * k] : (int16_t)0; v[C - 1][k] = k < left ? p[2 * k + 1] : (int16_t)0; }
    }
    return;
  }
#pragma unroll
  for (int c = 0; c < C; ++c) {
    const int16_t* p = s.p + (int64_t)c * s.n + i0;
    if (left >= 8 && (reinterpret_cast<uintptr_t>(p) & 15) == 0) {
      const int4 a = *reinterpret_cast<const int4*>(p);
      const int w[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
      for (int k = 0; k < 4; ++k) { v[c][2 * k] = (int16_t)(w[k] & 0xffff); v[c][2 * k + 1] = (int16_t)(w[k] >> 16); }
    } else {
#pragma unroll
      for (int k = 0; k < 8; ++k) v[c][k] = k < left ? p[k] : (int16_t)0;
    }
  }
}

// per block: sum(x) over all channels and sum(x^2) per channel in float64, x = float16(pcm) (:156, :1137-1139)
template <int C>
__global__ void __launch_bounds__(256) k_pcm_moments(const Pcm s, double* __restrict__ partials /* [blocks][3] */) {
  double sum = 0, q0 = 0, q1 = 0;
  const int64_t groups = (s.n + 7) / 8;
  for (int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; g < groups; g += (int64_t)gridDim.x * blockDim.x) {
    int16_t v[C][8];
    load_frames8<C>(s, 8 * g, v);               // frames past the end read as 0 and add nothing
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const double a = (double)(half_t)(float)v[0][k];
      sum += a; q0 += a * a;
      if (C == 2) { const double b = (double)(half_t)(float)v[C - 1][k]; sum += b; q1 += b * b; }
    }
  }
  __shared__ double red[3][256];
  red[0][threadIdx.x] = sum; red[1][threadIdx.x] = q0; red[2][threadIdx.x] = q1;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if ((int)threadIdx.x < w)
      for (int k = 0; k < 3; ++k) red[k][threadIdx.x] += red[k][threadIdx.x + w];
    __syncthreads();
  }
  if (threadIdx.x == 0)
    for (int k = 0; k < 3; ++k) partials[3 * blockIdx.x + k] = red[k][0];
}

struct Gain { int op[2]; double f[2]; };       // per channel: 0 keep, 1 divide by f, 2 multiply by f (:1144-1148)

// float16(pcm) (:156), then the float16-array-by-float64-scalar loudness gain, planar out
template <int C>
__global__ void __launch_bounds__(256) k_pcm_to_f16(const Pcm s, const Gain g, half_t* __restrict__ out, int64_t out_stride) {
  const int64_t grp = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t i0 = 8 * grp;
  if (i0 >= s.n) return;
  int16_t v[C][8];
  load_frames8<C>(s, i0, v);
#pragma unroll
  for (int c = 0; c < C; ++c) {
    half_t h[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      h[k] = (half_t)(float)v[c][k];
      if (g.op[c] == 1) h[k] = to_half((double)h[k] / g.f[c]);
      else if (g.op[c] == 2) h[k] = to_half((double)h[k] * g.f[c]);
    }
    half_t* o = out + (int64_t)c * out_stride + i0;
    if (s.n - i0 >= 8) *reinterpret_cast<int4*>(o) = *reinterpret_cast<const int4*>(h);
    else for (int k = 0; k < (int)(s.n - i0); ++k) o[k] = h[k];
  }
}

__global__ void __launch_bounds__(256) k_absmax(const half_t* __restrict__ x, int64_t n, int channels, int64_t stride,
                                                unsigned int* __restrict__ out) {
  unsigned int m = 0;
  const int64_t groups = (n + 7) / 8;
  for (int c = 0; c < channels; ++c) {
    const unsigned short* p = reinterpret_cast<const unsigned short*>(x) + (int64_t)c * stride;
    for (int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; g < groups; g += (int64_t)gridDim.x * blockDim.x) {
      if (n - 8 * g >= 8) {
        const int4 a = *reinterpret_cast<const int4*>(p + 8 * g);
        const unsigned int w[4] = {(unsigned)a.x, (unsigned)a.y, (unsigned)a.z, (unsigned)a.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const unsigned int lo = w[k] & 0x7fffu, hi = (w[k] >> 16) & 0x7fffu;      // |x| as ordered bits
          m = m > lo ? m : lo; m = m > hi ? m : hi;
        }
      } else {
        for (int64_t i = 8 * g; i < n; ++i) { const unsigned int b = p[i] & 0x7fffu; m = m > b ? m : b; }
      }
    }
  }
  __shared__ unsigned int red[256];
  red[threadIdx.x] = m; __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if ((int)threadIdx.x < w) red[threadIdx.x] = red[threadIdx.x] > red[threadIdx.x + w] ? red[threadIdx.x] : red[threadIdx.x + w];
    __syncthreads();
  }
  if (threadIdx.x == 0) atomicMax(out, red[0]);
}

// (:1153) video *= float16(32766 / max|video|) in float16 arithmetic, then (:136) int16, interleaved
template <int C>
__global__ void __launch_bounds__(256) k_finish(const half_t* __restrict__ x, int64_t n, int64_t stride,
                                                const unsigned int* __restrict__ peak_bits, int16_t* __restrict__ out) {
  const int64_t i0 = 8 * ((int64_t)blockIdx.x * blockDim.x + threadIdx.x);
  if (i0 >= n) return;
  const unsigned short pb = (unsigned short)*peak_bits;
  const half_t peak = *reinterpret_cast<const half_t*>(&pb);
  const half_t top = (half_t)32766.0f;                                  // 32768 in float16
  const half_t gain = (half_t)((float)top / (float)peak);
  const int cnt = n - i0 >= 8 ? 8 : (int)(n - i0);
  int16_t r[8 * C];
#pragma unroll
  for (int c = 0; c < C; ++c) {
    half_t h[8];
    const half_t* p = x + (int64_t)c * stride + i0;
    if (cnt == 8) *reinterpret_cast<int4*>(h) = *reinterpret_cast<const int4*>(p);
    else for (int k = 0; k < 8; ++k) h[k] = k < cnt ? p[k] : (half_t)0.0f;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const half_t v = (half_t)((float)h[k] * (float)gain);
      // numpy's float16 -> int16 cast goes through int32 and wraps: the reference's own peak sample
      // (32768 after the float16 rescale) comes out as -32768; same here
      const int32_t wide = (int32_t)(float)v;
      r[k * C + c] = (int16_t)(wide & 0xffff);
    }
  }
  int16_t* o = out + i0 * C;
  if (cnt == 8) {
#pragma unroll
    for (int q = 0; q < C; ++q) reinterpret_cast<int4*>(o)[q] = reinterpret_cast<const int4*>(r)[q];
  } else {
    for (int k = 0; k < cnt * C; ++k) o[k] = r[k];
  }
}

// ------------------------------------------------------------------------------------ resampling

constexpr int kFront = 24;     // rows after which the eliminated super-diagonal has reached its fixed point
constexpr int kFusedMinRows = 64;   // shorter splines take the sequential kernels, longer ones k_resample_tile

struct ResChunk {          // one block of <= 1e5 output points of one resampled interval (:234-243)
  int64_t out_abs;         // absolute video sample index of its first point
  int64_t first;           // index of its first point within the interval (k of the linspace)
  int32_t count;           // points in the block
  int32_t n;               // the spline runs through audio samples [b0, b0 + n)
  int64_t b0;
  int64_t coef_off;        // offset in doubles of its coefficients [channel][n]
  double start, step;      // np.linspace(x0, x1, num, endpoint=False): p_k = k * step + start
};

// knot j (0 .. n+2) of make_interp_spline(k=2) through x = b0 .. b0+n-1: end points tripled,
// mid-points in between with the first and last mid-point removed.
__host__ __device__ __forceinline__ double knot(int64_t j, int32_t n, int64_t b0) {
  if (j <= 2) return (double)b0;
  if (j >= n) return (double)(b0 + n - 1);
  return (double)(b0 + j - 3) + 1.5;
}

__host__ __device__ __forceinline__ int32_t knot_interval(double x, int32_t n, int64_t b0) {
  const double u = x - (double)b0;
  if (u < 1.5) return 2;
  const int64_t e = (int64_t)floor(u - 1.5) + 3;
  return (int32_t)(e < n - 1 ? e : n - 1);
}

// de Boor's recurrence for the three quadratic B-splines that are non-zero on knot interval ell
__host__ __device__ __forceinline__ void basis3(double x, int32_t ell, int32_t n, int64_t b0, double (&h)[3]) {
  h[0] = 1.0; h[1] = 0.0; h[2] = 0.0;
#pragma unroll
  for (int j = 1; j <= 2; ++j) {
    double hh0 = h[0], hh1 = h[1];
    h[0] = 0.0;
#pragma unroll
    for (int m = 1; m <= j; ++m) {
      const double xb = knot(ell + m, n, b0), xa = knot(ell + m - j, n, b0);
      if (xb == xa) { h[m] = 0.0; continue; }
      const double w = (m == 1 ? hh0 : hh1) / (xb - xa);
      h[m - 1] = h[m - 1] + w * (xb - x);
      h[m] = w * (x - xa);
    }
  }
}

// collocation row i: the spline basis at x = b0 + i restricted to columns i-1, i, i+1
__host__ __device__ __forceinline__ void colloc_row(int32_t i, int32_t n, int64_t b0, double& lo, double& di, double& up) {
  if (i >= 3 && i < n - 3) { lo = 0.125; di = 0.75; up = 0.125; return; }
  const double x = (double)(b0 + i);
  const int32_t ell = knot_interval(x, n, b0);
  double h[3]; basis3(x, ell, n, b0, h);
  lo = di = up = 0.0;
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    const int32_t col = ell - 2 + a;
    if (col == i - 1) lo = h[a]; else if (col == i) di = h[a]; else if (col == i + 1) up = h[a];
  }
}


__global__ void __launch_bounds__(64) k_spline_solve(const ResChunk* __restrict__ chunks, int n_chunks, int channels,
                                                     const half_t* __restrict__ audio, int64_t n_audio,
                                                     double* __restrict__ coef) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n_chunks * channels) return;
  const ResChunk ck = chunks[t / channels];
  const int ch = t % channels;
  const int32_t n = ck.n;
  if (n >= kFusedMinRows) return;                  // solved tile-wise by k_resample_tile
  const half_t* y = audio + (int64_t)ch * n_audio + ck.b0;
  double* c = coef + ck.coef_off + (int64_t)ch * n;
  // forward elimination; rows kFront .. n-4 are identical, so their pivot is computed once
  double cp_front[kFront];
  double tail[3] = {0.0, 0.0, 0.0};              // pivots of the last three rows
  double cp = 0.0, dp = 0.0, cp_mid = 0.0, w_mid = 0.0;
  for (int32_t i = 0; i < n; ++i) {
    double w;
    if (i >= kFront && i < n - 3) {
      w = w_mid; cp = cp_mid;
      dp = ((double)y[i] - 0.125 * dp) * w;
    } else {
      double lo, di, up; colloc_row(i, n, ck.b0, lo, di, up);
      w = 1.0 / (di - lo * cp);
      dp = ((double)y[i] - lo * dp) * w;
      cp = up * w;
      if (i < kFront) cp_front[i] = cp;
      if (i >= n - 3) tail[i - (n - 3)] = cp;
      if (i == kFront - 1) { cp_mid = 0.125 / (0.75 - 0.125 * cp); w_mid = 1.0 / (0.75 - 0.125 * cp); }
    }
    c[i] = dp;
  }
  // back substitution
  double next = c[n - 1];
  for (int32_t i = n - 2; i >= 0; --i) {
    double p;
    if (i >= n - 3) p = tail[i - (n - 3)];
    else if (i < kFront) p = cp_front[i];
    else p = cp_mid;
    next = c[i] - p * next;
    c[i] = next;
  }
}

__global__ void __launch_bounds__(256) k_spline_eval(const ResChunk* __restrict__ chunks, int channels,
                                                     const double* __restrict__ coef, half_t* __restrict__ video,
                                                     int64_t n_video) {
  const ResChunk ck = chunks[blockIdx.y];
  const int32_t k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= ck.count || ck.n >= kFusedMinRows) return;
  const double x = (double)(ck.first + k) * ck.step + ck.start;         // y = arange * step; y += start
  const bool inside = !(x < (double)ck.b0) && !(x > (double)(ck.b0 + ck.n - 1));
  int32_t ell = 2; double h[3] = {0, 0, 0};
  if (inside) { ell = knot_interval(x, ck.n, ck.b0); basis3(x, ell, ck.n, ck.b0, h); }
  for (int ch = 0; ch < channels; ++ch) {
    double v = 0.0;
    if (inside) {
      const double* c = coef + ck.coef_off + (int64_t)ch * ck.n + (ell - 2);
      v = h[0] * c[0];
      v = v + h[1] * c[1];
      v = v + h[2] * c[2];
    }
    video[(int64_t)ch * n_video + ck.out_abs + k] = to_half(v);
  }
}

// ---- fused tile kernel --------------------------------------------------------------------
// The collocation matrix is (1/8, 3/4, 1/8) away from the first/last three rows, so the solution
// at row i depends on the data at distance d with weight ~0.17^d.  A workgroup therefore solves a
// TILE of 4064 coefficient rows independently of the rest of the block: every thread runs the
// Thomas recurrences over its own 16 rows after a 20-row warm-up (error < 1e-15 relative; at the
// true ends of the spline the exact boundary rows are used instead), coefficients stay in LDS, and
// the same workgroup evaluates the output points that fall on its rows.  HBM traffic is the float16
// samples read once and the float16 result written once.
constexpr int kTileRows = 4064;          // coefficient rows solved per workgroup (+ kApron = 128 solver threads x kOwn)
#ifndef DA_RS_OWN
#define DA_RS_OWN 16
#endif
constexpr int kTileStep = kTileRows - 2; // consecutive tiles overlap by two rows (a point needs c[ell-2..ell])
constexpr int kOwn = DA_RS_OWN;          // rows per solver thread (16: all 256 threads solve; 32: half of them, 28 % fewer steps but 10 % slower)
constexpr int kWarm = 20;                // warm-up rows
constexpr int kApron = 32;               // forward-pass rows beyond the tile (warm-up of the backward pass)
constexpr int kMaxTiles = 28;            // ceil((1.1e5 + 4) / 4062): the rate is within +-10 % (:33)

struct SplineRows {            // tabulated on the host with the same recurrences (n >= kFusedMinRows)
  double lo_front[kFront], w_front[kFront], cp_front[kFront];
  double lo_tail[3], w_tail[3], cp_tail[3];
  double w_mid, cp_mid;
};

__device__ __forceinline__ int pad_y(int r) { return r + 2 * (r / kOwn); }     // float16 index, conflict-free per owner
__device__ __forceinline__ int pad_c(int r) { return r + (r / kOwn); }         // float64 index

__global__ void __launch_bounds__(256) k_resample_tile(const ResChunk* __restrict__ chunks, const half_t* __restrict__ audio,
                                                       int64_t n_audio, half_t* __restrict__ video, int64_t n_video,
                                                       const SplineRows R) {
  const ResChunk ck = chunks[blockIdx.y];
  const int32_t n = ck.n;
  const int32_t R0 = (int32_t)blockIdx.x * kTileStep - 2;            // first coefficient row of the tile
  if (n < kFusedMinRows || R0 + 2 >= n) return;
  const int ch = blockIdx.z;
  const double w_mid = R.w_mid, cp_mid = R.cp_mid;
  const double nq_mid = -0.125 * w_mid, ncp_mid = -cp_mid;
  constexpr int kRowsY = kTileRows + kApron + kWarm;                  // rows [R0 - kWarm, R0 + 4128)
  __shared__ half_t ys[kRowsY + 2 * (kRowsY / 16) + 8];
  __shared__ double cs[kTileRows + kApron + (kTileRows + kApron) / 16 + 4];
  const half_t* y = audio + (int64_t)ch * n_audio + ck.b0;
  const int32_t ybase = R0 - kWarm;
  for (int r = threadIdx.x; r < kRowsY; r += 256) {
    const int32_t row = ybase + r;
    ys[pad_y(r)] = (row >= 0 && row < n) ? y[row] : (half_t)0.0f;
  }
  __syncthreads();
#ifndef DA_DBG_RS_NOSOLVE
  // ---- forward elimination: task j owns rows [R0 + 16 j, +16)
  for (int task = threadIdx.x; task < (kTileRows + kApron) / kOwn; task += 256) {
    const int32_t lo_row = R0 + kOwn * task;
    int32_t first = lo_row < 0 ? 0 : lo_row;
    const int32_t last = lo_row + kOwn < n ? lo_row + kOwn : n;     // exclusive
    if (first >= last) continue;
    int32_t s0 = first - kWarm;
    if (s0 < kFront) s0 = 0;                                         // reach the true first rows instead of guessing
    double dp = 0.0;
    int32_t i = s0;
    if (s0 == 0) {                                                   // the spline's true first rows (first tile only)
      const int32_t fe = kFront < last ? kFront : last;
      for (; i < fe; ++i) {
        dp = ((double)ys[pad_y(i - ybase)] - R.lo_front[i] * dp) * R.w_front[i];
        if (i >= first) cs[pad_c(i - R0)] = dp;
      }
    }
    const int32_t ie = last < n - 3 ? last : n - 3;
    for (; i + 8 <= ie; i += 8) {                                    // uniform rows: constants in registers, LDS only;
      double yv[8];                                                  // the reads go out together ahead of the dependent chain
#pragma unroll
      for (int e = 0; e < 8; ++e) yv[e] = (double)ys[pad_y(i + e - ybase)] * w_mid;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        dp = fma(nq_mid, dp, yv[e]);                                 // (y - dp/8) w as ONE dependent operation
        if (i + e >= first) cs[pad_c(i + e - R0)] = dp;
      }
    }
    for (; i < ie; ++i) {
      dp = fma(nq_mid, dp, (double)ys[pad_y(i - ybase)] * w_mid);
      if (i >= first) cs[pad_c(i - R0)] = dp;
    }
    for (; i < last; ++i) {                                          // the true last three rows (last tile only)
      dp = ((double)ys[pad_y(i - ybase)] - R.lo_tail[i - (n - 3)] * dp) * R.w_tail[i - (n - 3)];
      if (i >= first) cs[pad_c(i - R0)] = dp;
    }
  }
  __syncthreads();
  // ---- back substitution over the tile's own rows (dp of the rows to the right is in LDS)
  double cown[kOwn];
  {
    const int task = threadIdx.x;
    const int32_t lo_row = R0 + kOwn * task;
    const int32_t first = lo_row < 0 ? 0 : lo_row;
    const int32_t last = task >= kTileRows / kOwn ? first : (lo_row + kOwn < n ? lo_row + kOwn : n);   // apron rows are not finished
    if (first < last) {
      int32_t e = last + kWarm;
      if (e > n) e = n;
      if (e > R0 + kTileRows + kApron) e = R0 + kTileRows + kApron;
      double next = 0.0;
      auto pivot = [&](int32_t i) -> double {
        if (i >= kFront && i < n - 3) return cp_mid;                 // register constant; the tables only at the true ends
        return i < kFront ? R.cp_front[i] : R.cp_tail[i - (n - 3)];
      };
      int32_t i = e - 1;
      for (; i >= last && (i >= n - 3 || i - 3 < last); --i) next = cs[pad_c(i - R0)] - pivot(i) * next;   // true last rows / remainder
      for (; i - 3 >= last && i - 3 >= kFront; i -= 4) {                                           // warm-up, uniform rows
        double d4[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) d4[q] = cs[pad_c(i - q - R0)];
#pragma unroll
        for (int q = 0; q < 4; ++q) next = fma(ncp_mid, next, d4[q]);
      }
      for (; i >= last; --i) next = cs[pad_c(i - R0)] - pivot(i) * next;
#pragma unroll
      for (int k = 0; k < kOwn; ++k) {                               // this thread's own dp values, read together
        const int32_t r = lo_row + k;
        cown[k] = (r >= first && r < last) ? cs[pad_c(r - R0)] : 0.0;
      }
#pragma unroll
      for (int k = kOwn - 1; k >= 0; --k) {
        const int32_t r = lo_row + k;
        if (r >= first && r < last) { next = fma(-pivot(r), next, cown[k]); cown[k] = next; }
      }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < kOwn; ++k) {
      const int32_t i = lo_row + k;
      if (i >= first && i < last) cs[pad_c(i - R0)] = cown[k];
    }
  }
  __syncthreads();
#endif
  // ---- evaluate the points whose knot interval ell lies in [E0, E1)
  const int32_t E0 = R0 + 2 < 2 ? 2 : R0 + 2;
  const int32_t E1 = R0 + kTileRows < n ? R0 + kTileRows : n;
  auto x_of = [&](int64_t k) { return (double)(ck.first + k) * ck.step + ck.start; };   // y = arange * step; y += start
  auto count_below = [&](double X) -> int64_t {                      // number of points with x_k < X
    double est = ceil((X - ck.start) / ck.step) - (double)ck.first;
    int64_t k = est < 0 ? 0 : (est > (double)ck.count ? (int64_t)ck.count : (int64_t)est);
    while (k > 0 && !(x_of(k - 1) < X)) --k;
    while (k < ck.count && x_of(k) < X) ++k;
    return k;
  };
  const int64_t k0 = E0 <= 2 ? 0 : count_below((double)ck.b0 + (double)E0 - 1.5);
  const int64_t k1 = E1 >= n ? (int64_t)ck.count : count_below((double)ck.b0 + (double)E1 - 1.5);
  half_t* out = video + (int64_t)ch * n_video + ck.out_abs;
#ifdef DA_DBG_RS_NOEVAL
  if (k0 >= 0) { if (threadIdx.x == 0 && k0 < k1) out[k0] = (half_t)(float)cs[pad_c(3)]; return; }
#endif
  for (int64_t k = k0 + threadIdx.x; k < k1; k += 256) {
    const double x = x_of(k);
    double v = 0.0;
    if (!(x < (double)ck.b0) && !(x > (double)(ck.b0 + n - 1))) {
      const int32_t ell = knot_interval(x, n, ck.b0);
      double h[3];
      if (ell >= 5 && ell <= n - 4) {
        // interior knots are one apart: de Boor's divisions are by 1 and 2 (exact as multiplications)
        const double t0 = (double)(ck.b0 + ell - 3) + 1.5;           // knot(ell)
        const double a = x - t0, b = (t0 + 1.0) - x;                 // x - t[ell], t[ell+1] - x
        const double w1 = 0.5 * b, w2 = 0.5 * a;                     // h1 / (t[ell+1]-t[ell-1]), h2 / (t[ell+2]-t[ell])
        h[0] = w1 * b;
        h[1] = w1 * (x - (t0 - 1.0));
        h[1] = h[1] + w2 * ((t0 + 2.0) - x);
        h[2] = w2 * a;
      } else {
        basis3(x, ell, n, ck.b0, h);
      }
      const int r = ell - 2 - R0;
      v = h[0] * cs[pad_c(r)];
      v = v + h[1] * cs[pad_c(r + 1)];
      v = v + h[2] * cs[pad_c(r + 2)];
    }
    out[k] = to_half(v);
  }
}

// ------------------------------------------------------------------------------------ stretching

// The reference forms every sliding 512-sum as a difference of one float64 running sum over the
// chunk (np.cumsum, strictly sequential; :275-277, :283-284).  In quiet passages that running sum
// absorbs small products entirely, correlations come out as exactly 1.0 and np.argmax picks the
// first of many ties -- so the jump positions depend on the rounding of that particular sum.
// These kernels therefore keep the reference's order of operations: one thread walks one chunk
// sequentially with two running sums 512 elements apart (cs[p+511] and cs[p-1], both accumulated
// in the same order as the reference's single sum, hence bit-identical to it).  Parallelism is
// over (chunk, lag): 10-482 lags x one chunk per 25 088 samples.

struct SegDesc {             // one stretched interval
  int64_t in_off;            // x0: first audio sample of the interval
  int64_t n_in, n_out, total, n_windows;
  int n_lags, lag_off;
  int64_t table_off;         // offset into where/loss  (n_windows * n_lags entries)
  int64_t back_off;          // offset into the back-pointer array (n_windows * 3073 entries)
  int64_t plan_off;          // offset into the copy plan arrays (n_windows + 2 entries)
};

struct CorrChunk {            // a chunk of <= 57 windows (:253-270), all intervals in one list
  int64_t begin, end;         // samples of its interval it is computed from
  int64_t rms_off;            // offset of its rms[0 .. end-begin-511) in the rms array
  int32_t w_lo, w_hi;         // local windows it hands out
  int32_t seg;                // which stretched interval
  int32_t pad_;
};

__device__ __forceinline__ float lag_product(const half_t* __restrict__ s, int64_t t, int lag, int channels, int64_t ch_stride) {
  float r = (float)s[t + lag] * (float)s[t];
  if (channels == 2) r = r + (float)s[ch_stride + t + lag] * (float)s[ch_stride + t];
  return r;
}

// One wavefront per running sum.  Element block m = products [512 m, 512 m + 512): all lanes
// fetch them (coalesced) into LDS, lane 0 extends the float64 running sum over them in order and
// leaves cs[t] in a ring of the last 2048 elements; the lanes then share the per-position work.
constexpr int kRing = 2048;

__device__ __forceinline__ void extend_running_sum(const half_t* __restrict__ s, int64_t n_prod, int lag, int channels,
                                                   int64_t ch_stride, int64_t m, float* stage, double* ring, double& cs) {
  const int lane = threadIdx.x;
  const int64_t t0 = m * kSW;
#pragma unroll
  for (int k = 0; k < kSW / 64; ++k) {
    const int64_t t = t0 + lane + 64 * k;
    stage[lane + 64 * k] = t < n_prod ? lag_product(s, t, lag, channels, ch_stride) : 0.f;
  }
  __syncthreads();                 // one-wavefront workgroup: orders the LDS traffic, costs nothing
  if (lane == 0) {
    double acc = cs;
    const int64_t cnt = n_prod - t0 < kSW ? n_prod - t0 : kSW;
    int64_t k = 0;
    for (; k + 16 <= cnt; k += 16) {               // LDS reads up front, then the dependent chain of adds
      float v[16]; double c[16];
#pragma unroll
      for (int e = 0; e < 16; ++e) v[e] = stage[k + e];
#pragma unroll
      for (int e = 0; e < 16; ++e) { acc += (double)v[e]; c[e] = acc; }
#pragma unroll
      for (int e = 0; e < 16; ++e) ring[(t0 + k + e) & (kRing - 1)] = c[e];
    }
    for (; k < cnt; ++k) { acc += (double)stage[k]; ring[(t0 + k) & (kRing - 1)] = acc; }
    cs = acc;
  }
  __syncthreads();
}

// wave per chunk: rms[p] = sqrt(E[p] + eps), E[p] = sum_{k<512} power[p+k], eps = 1e-4 max(1, max E)  (:272-279)
__global__ void __launch_bounds__(64) k_chunk_rms(const half_t* __restrict__ audio, int channels, int64_t ch_stride,
                                                  const SegDesc* __restrict__ segs, const CorrChunk* __restrict__ chunks,
                                                  double* __restrict__ rms_all, double* __restrict__ eps_all) {
  __shared__ float stage[kSW];
  __shared__ double ring[kRing];
  const CorrChunk ck = chunks[blockIdx.x];
  const half_t* s = audio + segs[ck.seg].in_off + ck.begin;
  const int64_t L = ck.end - ck.begin;
  const int64_t P = L - kSW + 1;
  double* rms = rms_all + ck.rms_off;
  const int lane = threadIdx.x;
  double cs = 0.0, mx = 0.0;
  const int64_t n_blocks = (L + kSW - 1) / kSW;
  for (int64_t m = 0; m < n_blocks; ++m) {
    extend_running_sum(s, L, 0, channels, ch_stride, m, stage, ring, cs);
    // positions whose window ends inside this element block: p + 511 in [512 m, 512 m + 512)
    for (int k = lane; k < kSW; k += 64) {
      const int64_t p = m * kSW + k - (kSW - 1);
      if (p >= 0 && p < P) {
        const double hi = ring[(p + kSW - 1) & (kRing - 1)];
        const double e = p ? hi - ring[(p - 1) & (kRing - 1)] : hi;
        rms[p] = e;
        mx = e > mx ? e : mx;
      }
    }
  }
  for (int off = 32; off > 0; off >>= 1) { const double o = __shfl_xor(mx, off); mx = o > mx ? o : mx; }
  const double eps = 1e-4 * (mx > 1.0 ? mx : 1.0);
  if (lane == 0) eps_all[blockIdx.x] = eps;
  __threadfence();
  __syncthreads();
  for (int64_t p = lane; p < P; p += 64) rms[p] = sqrt(rms[p] + eps);
}

__global__ void k_fill_table(int16_t* __restrict__ where, double* __restrict__ loss, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) { where[i] = 0; loss[i] = INFINITY; }        // np.argmax of an all -inf column is 0, its loss 1 - (-inf)
}

// wave per (chunk, lag): Pearson correlation of the window at q with the window at q + lag for
// every q of the chunk, arg-max and 1 - max per 512-position window (:280-296, :321-322).
__global__ void __launch_bounds__(64) k_lag_table(const half_t* __restrict__ audio, int channels, int64_t ch_stride,
                                                  const SegDesc* __restrict__ segs, const CorrChunk* __restrict__ chunks,
                                                  const int32_t* __restrict__ lags_all, const double* __restrict__ rms_all,
                                                  const double* __restrict__ eps_all, int16_t* __restrict__ where_all,
                                                  double* __restrict__ loss_all) {
  __shared__ float stage[kSW];
  __shared__ double ring[kRing];
  const CorrChunk ck = chunks[blockIdx.x];
  const SegDesc sd = segs[ck.seg];
  const int j = blockIdx.y;
  if (j >= sd.n_lags) return;
  const int lag = lags_all[sd.lag_off + j];
  const bool backwards = sd.total > 0;
  const double eps = eps_all[blockIdx.x];
  const half_t* s = audio + sd.in_off + ck.begin;
  const double* rms = rms_all + ck.rms_off;
  const int64_t L = ck.end - ck.begin;
  const int64_t P = L - kSW + 1;
  const int64_t Q = P - lag;                               // window starts q with q + lag still inside
  if (Q <= 0) return;
  const int64_t n_prod = L - lag;                          // lag products of the chunk
  const int64_t w0 = ck.begin / kSW;
  const int lane = threadIdx.x;
  const int off = backwards ? lag : 0;                     // row p of the reference's matrix is q + off
  const int64_t n_win = (Q + off + kSW - 1) / kSW;         // windows that contain at least one row
  const int64_t n_blocks = (n_prod + kSW - 1) / kSW;
  double cs = 0.0;
  int64_t made = 0;                                        // element blocks already summed
  for (int64_t w = 0; w < n_win; ++w) {
    // rows p in [512 w, 512 w + 512) need running sums up to element p - off + 511 < 512 (w + 2)
    while (made < n_blocks && made <= w + 1) { extend_running_sum(s, n_prod, lag, channels, ch_stride, made, stage, ring, cs); ++made; }
    const bool wanted = w >= ck.w_lo && w < ck.w_hi && w0 + w < sd.n_windows;
    if (!wanted) continue;
    double best = -INFINITY; int best_u = kSW;
#pragma unroll
    for (int k = 0; k < kSW / 64; ++k) {
      const int u = lane + 64 * k;                         // ascending per lane: first maximum kept by '>'
      const int64_t q = w * kSW + u - off;
      if (q >= 0 && q < Q) {
        const double hi = ring[(q + kSW - 1) & (kRing - 1)];
        const double dots = (q ? hi - ring[(q - 1) & (kRing - 1)] : hi) + eps;
        const double ra = rms[q], rb = rms[q + lag];
        const double corr = backwards ? (dots / ra) / rb : (dots / rb) / ra;   // (dots / rms[other window]) / rms[p]
        if (corr > best) { best = corr; best_u = u; }
      }
    }
    for (int sh = 32; sh > 0; sh >>= 1) {
      const double ob = __shfl_xor(best, sh); const int ou = __shfl_xor(best_u, sh);
      if (ob > best || (ob == best && ou < best_u)) { best = ob; best_u = ou; }
    }
    if (lane == 0 && best != -INFINITY) {
      where_all[sd.table_off + (w0 + w) * sd.n_lags + j] = (int16_t)best_u;
      loss_all[sd.table_off + (w0 + w) * sd.n_lags + j] = 1.0 - best;
    }
  }
}

__device__ __forceinline__ int64_t floordiv(int64_t a, int64_t b) {     // Python's // for b > 0
  int64_t q = a / b;
  if ((a % b != 0) && (a < 0)) --q;
  return q;
}
__device__ __forceinline__ int64_t offset_at(int64_t total, int64_t nw, int64_t w) {       // (:310-311)
  const int64_t c = w < 0 ? 0 : (w > nw - 1 ? nw - 1 : w);
  return floordiv(total * c, nw - 1);
}
__device__ __forceinline__ int64_t offset_step(int64_t total, int64_t nw, int64_t w) {     // (:316-317)
  const int64_t d = offset_at(total, nw, w) - offset_at(total, nw, w - 1);
  return d < 0 ? -d : d;
}

// One workgroup per stretched interval.  LDS: cost history [3][3073] f64, loss row, lags.
__global__ void __launch_bounds__(1024) k_viterbi(const SegDesc* __restrict__ segs, const int32_t* __restrict__ lags_all,
                                                  const int16_t* __restrict__ where_all, const double* __restrict__ loss_all,
                                                  int16_t* __restrict__ back_all, int64_t* __restrict__ plan_in,
                                                  int64_t* __restrict__ plan_out, int64_t* __restrict__ sched,
                                                  int32_t* __restrict__ counts) {
  extern __shared__ double lds[];
  // cost history: three rows of 3073 drift states, each with guard zones of +inf on both sides so
  // that the jump look-ups h2[d + two - lag] need no range checks (a jump from outside the drift
  // window costs +inf, exactly the reference's untouched np.inf entries)
  double* hist = lds;                          // [3][kRow]
  double* lrow = lds + 3 * kRow;               // [2][kMaxLags]: loss row of this window / the next one
  int32_t* lags = reinterpret_cast<int32_t*>(lrow + 2 * kMaxLags);   // [kMaxLags]
  __shared__ int64_t sstep[2];
  const SegDesc sd = segs[blockIdx.x];
  const int J = sd.n_lags;
  const int64_t nw = sd.n_windows;
  const double* loss = loss_all + sd.table_off;
  const int16_t* where = where_all + sd.table_off;
  int16_t* back = back_all + sd.back_off;
  const int tid = threadIdx.x;
  for (int d = tid; d < 3 * kRow; d += blockDim.x) hist[d] = INFINITY;
  if (tid < J) { lags[tid] = lags_all[sd.lag_off + tid]; lrow[tid] = loss[tid]; }
  if (tid == 0) sstep[0] = offset_step(sd.total, nw, 0);
  __syncthreads();
  if (tid == 0) { hist[1 * kRow + kGuardLo + kMaxDrift] = 0.0; hist[2 * kRow + kGuardLo + kMaxDrift] = 0.0; }      // (:320)
  __syncthreads();
  // drift states of this thread: tid, tid + 1024, tid + 2048; state 3072 is thread 0's extra
  constexpr int kSlots = 3;
  int64_t prev_step = 0;
  bool overflow = false;
  for (int64_t w = 0; w < nw; ++w) {
    const int cur = (int)(w & 1);
    // the next window's loss row and drift step are fetched while this window is processed
    double nxt = 0.0; int64_t nstep = 0;
    if (tid < J && w + 1 < nw) nxt = loss[(w + 1) * J + tid];
    if (tid == 0) nstep = offset_step(sd.total, nw, w + 1);
    const int step = (int)sstep[cur];
    const int two = step + (int)prev_step;
    if (two > kGuardHi) { overflow = true; break; }      // uniform; cannot happen for |1 - rate| <= 0.1
    const double* h1 = hist + ((w + 2) % 3) * kRow + kGuardLo;       // (w-1) % 3
    const double* h2 = hist + ((w + 1) % 3) * kRow + kGuardLo;       // (w-2) % 3
    double* hw = hist + (w % 3) * kRow + kGuardLo;
    const double* lr = lrow + cur * kMaxLags;
    double best[kSlots]; int pick[kSlots];
#pragma unroll
    for (int m = 0; m < kSlots; ++m) { best[m] = h1[tid + 1024 * m + step]; pick[m] = 0; }   // no jump (:333-334); past the end: +inf guard
#pragma unroll 2
    for (int k = 0; k < J; ++k) {
      const int lag = lags[k];
      const double lk = lr[k];
      const double* src = h2 + (two - lag);                                // (:335-342)
      double v[kSlots];
#pragma unroll
      for (int m = 0; m < kSlots; ++m) v[m] = src[tid + 1024 * m] + lk;
#pragma unroll
      for (int m = 0; m < kSlots; ++m) {
        // the destination slice starts at `lag` (< 512): only the first slot can lie below it
        const bool take = m == 0 ? ((tid >= lag) & (v[m] < best[m])) : (v[m] < best[m]);
        best[m] = take ? v[m] : best[m];
        pick[m] = take ? k + 1 : pick[m];
      }
    }
#pragma unroll
    for (int m = 0; m < kSlots; ++m) {
      const int d = tid + 1024 * m;
      hw[d] = best[m]; back[w * kND + d] = (int16_t)pick[m];
    }
    if (tid == 0) {
      const int d = kND - 1;
      double b = h1[d + step];
      int pk = 0;
      for (int k = 0; k < J; ++k) {
        const int lag = lags[k];
        const double v = h2[d + two - lag] + lr[k];
        if (d >= lag && v < b) { b = v; pk = k + 1; }
      }
      hw[d] = b; back[w * kND + d] = (int16_t)pk;
    }
    if (tid < J) lrow[(cur ^ 1) * kMaxLags + tid] = nxt;
    if (tid == 0) sstep[cur ^ 1] = nstep;
    prev_step = step;
    __syncthreads();
  }
  __threadfence();
  __syncthreads();
  if (threadIdx.x != 0) return;
  // back-track (:346-362); the schedule is written backwards into sched, then reversed in place
  int64_t* sc = sched + 2 * sd.plan_off;
  int64_t drift = kMaxDrift;
  int32_t K = 0;
  bool skip = false, bad = false;
  for (int64_t w = nw - 1; w >= 0; --w) {
    drift += offset_step(sd.total, nw, w + 1);
    if (skip) { skip = false; continue; }
    if (drift < 0 || drift >= kND) { bad = true; break; }
    const int k = (int)back[w * kND + drift] - 1;
    if (k < 0) continue;
    const int lag = lags[k];
    sc[2 * K] = w * kSW + (int64_t)where[w * J + k];
    sc[2 * K + 1] = sd.total > 0 ? -(int64_t)lag : (int64_t)lag;          // (:366-367)
    drift -= lag;
    skip = true;
    ++K;
  }
  for (int32_t i = 0; i < K / 2; ++i) {
    const int32_t o = K - 1 - i;
    const int64_t a0 = sc[2 * i], a1 = sc[2 * i + 1];
    sc[2 * i] = sc[2 * o]; sc[2 * i + 1] = sc[2 * o + 1];
    sc[2 * o] = a0; sc[2 * o + 1] = a1;
  }
  // copy plan (:370-376): run m copies input [in[m], ...) to output [out[m], out[m+1])
  int64_t* pin = plan_in + sd.plan_off;
  int64_t* pout = plan_out + sd.plan_off;
  int64_t start = 0, ostart = 0;
  for (int32_t m = 0; m <= K; ++m) {
    const int64_t end = m < K ? sc[2 * m] : sd.n_in;
    pin[m] = start; pout[m] = ostart;
    ostart += end - start;
    if (m < K) start = sc[2 * m] + sc[2 * m + 1];
  }
  pout[K + 1] = ostart;
  counts[2 * blockIdx.x] = K;
  counts[2 * blockIdx.x + 1] = (bad || overflow) ? 1 : 0;
}

struct SpliceArgs {
  const half_t* seg; int64_t n_in; int64_t in_stride;      // audio + x0, channel stride
  half_t* out; int64_t n_out; int64_t out_stride;          // video + y0
  int channels;
  const int64_t* pin; const int64_t* pout; const int32_t* count;   // copy plan of this interval
  const double* rise; const double* fall;                  // hann(1025)[:512], [512:1024]
};

__global__ void __launch_bounds__(256) k_splice(const SpliceArgs a) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int32_t M = a.count[0] + 1;                        // runs
  if (t >= a.n_out || t >= a.pout[M]) return;
  int32_t lo = 0, hi = M - 1;                              // largest m with pout[m] <= t
  while (lo < hi) { const int32_t mid = (lo + hi + 1) >> 1; if (a.pout[mid] <= t) lo = mid; else hi = mid - 1; }
  const int32_t m = lo;
  const int64_t u = t - a.pout[m];
  for (int ch = 0; ch < a.channels; ++ch) {
    const half_t* s = a.seg + (int64_t)ch * a.in_stride;
    half_t val;
    if (u >= kSW) {
      const int64_t src = a.pin[m] + u;                    // out[oa+W : ob+W] = seg[a+W : b+W]   (:385)
      if (src >= a.n_in) continue;
      val = s[src];
    } else {
      // the 512 samples after a jump: what the previous run left there, faded out, plus this run faded in
      int32_t m0 = m;
      while (m0 > 0 && t - a.pout[m0 - 1] < kSW) --m0;
      const int64_t bsrc = m0 == 0 ? t : a.pin[m0 - 1] + (t - a.pout[m0 - 1]);
      half_t acc = bsrc < a.n_in ? s[bsrc] : (half_t)0.0f;
      for (int32_t mm = m0; mm <= m; ++mm) {
        const int64_t uu = t - a.pout[mm];
        const int64_t src = a.pin[mm] + uu;
        acc = to_half((double)acc * a.fall[uu]);                                             // (:383)
        const double add = (src < a.n_in ? (double)s[src] : 0.0) * a.rise[uu];
        acc = to_half((double)acc + add);                                                    // (:384)
      }
      val = acc;
    }
    a.out[(int64_t)ch * a.out_stride + t] = val;
  }
}

}  // namespace

// ------------------------------------------------------------------------------------ host side

struct StretchState {
  DBuf res_chunks, coef;
  DBuf energy, corr_chunks, eps, lags, where, loss, back, segs, plan_in, plan_out, sched, counts;
  DBuf hann, partials, peak;
  bool hann_ready = false;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  std::vector<std::vector<int64_t>> schedules;       // per stretched interval: idx0, dist0, idx1, dist1, ...
};

StretchState* stretch_create() {
  StretchState* s = new StretchState();
  (void)hipEventCreate(&s->e0); (void)hipEventCreate(&s->e1);
  return s;
}

void stretch_destroy(StretchState* s) {
  if (!s) return;
  DBuf* all[] = {&s->res_chunks, &s->coef, &s->energy, &s->corr_chunks, &s->eps, &s->lags, &s->where, &s->loss,
                 &s->back, &s->segs, &s->plan_in, &s->plan_out, &s->sched, &s->counts, &s->hann, &s->partials, &s->peak};
  for (DBuf* b : all) b->release();
  if (s->e0) (void)hipEventDestroy(s->e0);
  if (s->e1) (void)hipEventDestroy(s->e1);
  delete s;
}

const std::vector<int64_t>* stretch_schedule(const StretchState* s, int k) {
  if (!s || k < 0 || k >= (int)s->schedules.size()) return nullptr;
  return &s->schedules[k];
}
int stretch_schedule_count(const StretchState* s) { return s ? (int)s->schedules.size() : 0; }

namespace {

int sfail(std::string& err, int code, const char* fmt, ...) {
  char buf[512];
  va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof buf, fmt, ap); va_end(ap);
  err = buf;
  return code;
}

#define S_TRY(call)                                                                              \
  do {                                                                                           \
    hipError_t e_ = (call);                                                                      \
    if (e_ != hipSuccess)                                                                        \
      return sfail(err, DA_ERR_DEVICE, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
  } while (0)

struct Interval { int kind; int64_t x0, x1, y0, y1; };     // kind 0 skip, 1 resample, 2 stretch

std::vector<int32_t> lag_list(int64_t total) {              // (:303-308)
  static const int32_t base[10] = {506, 451, 284, 410, 480, 379, 308, 430, 265, 494};
  const int64_t mag = total < 0 ? -total : total;
  std::vector<int32_t> l;
  if (mag >= 10000) { l.assign(base, base + 10); return l; }
  if (mag > 1000) {
    l.assign(base, base + 10);
    for (int b = 0; b < 8; ++b) l.push_back(kMinOffset + (1 << b) - 1);
    return l;
  }
  for (int v = kMinOffset; v < kSW; ++v) l.push_back(v);
  return l;
}

// appends the chunks of one interval of n samples (:253-270) to the flat list
void corr_chunks(int64_t n, int32_t seg, int64_t& rms_total, std::vector<CorrChunk>& out) {
  const double limit = (kCached + 2) * 1.1 * kSW;
  int64_t begin = 0;
  int32_t lo = 0;
  while (true) {
    const bool last = (double)(n - begin) <= limit;
    CorrChunk c{};
    c.begin = begin; c.end = last ? n : begin + (int64_t)(kCached + 1) * kSW;
    c.rms_off = rms_total; c.w_lo = lo; c.w_hi = last ? (int32_t)((n - begin) / kSW) : kCached;
    c.seg = seg;
    rms_total += c.end - c.begin - kSW + 1;
    out.push_back(c);
    if (last) return;
    begin += (int64_t)(kCached - 1) * kSW;
    lo = 1;
  }
}

SplineRows spline_rows() {
  // any n >= kFusedMinRows gives the same first kFront and last three rows
  const int32_t n = 4 * kFront; const int64_t b0 = 0;
  SplineRows R{};
  double cp = 0.0;
  for (int i = 0; i < kFront; ++i) {
    double lo, di, up; colloc_row(i, n, b0, lo, di, up);
    const double w = 1.0 / (di - lo * cp);
    cp = up * w;
    R.lo_front[i] = lo; R.w_front[i] = w; R.cp_front[i] = cp;
  }
  for (int it = 0; it < 64; ++it) cp = 0.125 / (0.75 - 0.125 * cp);          // the fixed point
  R.cp_mid = cp; R.w_mid = 1.0 / (0.75 - 0.125 * cp);
  for (int t = 0; t < 3; ++t) {
    double lo, di, up; colloc_row(n - 3 + t, n, b0, lo, di, up);
    const double w = 1.0 / (di - lo * cp);
    cp = up * w;
    R.lo_tail[t] = lo; R.w_tail[t] = w; R.cp_tail[t] = cp;
  }
  return R;
}

float elapsed(hipEvent_t a, hipEvent_t b) { float ms = 0.f; (void)hipEventElapsedTime(&ms, a, b); return ms; }

int ensure_hann(StretchState* s, hipStream_t stream, std::string& err) {
  if (s->hann_ready) return 0;
  // scipy.signal.windows.hann(1025): 0.5 + 0.5 cos(fac), fac = linspace(-pi, pi, 1025)
  const double pi = 3.141592653589793;
  std::vector<double> w(2 * kSW);
  const double step = (pi - (-pi)) / (2 * kSW);
  for (int k = 0; k < 2 * kSW; ++k) w[k] = 0.5 + 0.5 * std::cos((double)k * step + (-pi));
  S_TRY(s->hann.ensure(sizeof(double) * 2 * kSW));
  S_TRY(hipMemcpyAsync(s->hann.p, w.data(), sizeof(double) * 2 * kSW, hipMemcpyHostToDevice, stream));
  S_TRY(da::stream_wait(stream));
  s->hann_ready = true;
  return 0;
}

}  // namespace

int64_t stretch_channel_stride(int64_t n) { return (n + 63) / 64 * 64; }

int stretch_prepare(StretchState* s, hipStream_t stream, const int16_t* d_pcm_video, int64_t n_video, int planar_video,
                    const int16_t* d_pcm_audio, int64_t n_audio, int planar_audio, int channels, uint16_t* d_video,
                    uint16_t* d_audio, double* factors, std::string& err) {
  const int blocks = 2048;
  S_TRY(s->partials.ensure(sizeof(double) * 3 * blocks * 2));
  double* pv = s->partials.as<double>();
  double* pa = pv + 3 * blocks;
  const Pcm sv{d_pcm_video, n_video, planar_video}, sa{d_pcm_audio, n_audio, planar_audio};
  if (channels == 2) {
    hipLaunchKernelGGL(k_pcm_moments<2>, dim3(blocks), dim3(256), 0, stream, sv, pv);
    hipLaunchKernelGGL(k_pcm_moments<2>, dim3(blocks), dim3(256), 0, stream, sa, pa);
  } else {
    hipLaunchKernelGGL(k_pcm_moments<1>, dim3(blocks), dim3(256), 0, stream, sv, pv);
    hipLaunchKernelGGL(k_pcm_moments<1>, dim3(blocks), dim3(256), 0, stream, sa, pa);
  }
  S_TRY(hipGetLastError());
  std::vector<double> h(3 * blocks * 2);
  S_TRY(hipMemcpyAsync(h.data(), pv, sizeof(double) * h.size(), hipMemcpyDeviceToHost, stream));
  S_TRY(da::stream_wait(stream));
  auto spread = [&](const double* p, int64_t n, double* out) {           // low_ram_std (:1137-1139)
    double sum = 0, q[2] = {0, 0};
    for (int b = 0; b < blocks; ++b) { sum += p[3 * b]; q[0] += p[3 * b + 1]; q[1] += p[3 * b + 2]; }
    const double cnt = (double)n * channels;
    const double avg = sum / cnt;
    for (int c = 0; c < channels; ++c) out[c] = std::sqrt(q[c] / cnt - avg * avg);
  };
  double dv[2], da_[2];
  spread(h.data(), n_video, dv); spread(h.data() + 3 * blocks, n_audio, da_);
  Gain gv{}, ga{};
  for (int c = 0; c < channels; ++c) {
    const double f = dv[c] / da_[c];
    factors[c] = f;
    if (f > 1) { gv.op[c] = 1; gv.f[c] = f; }                            // (:1144-1148): only the louder track is scaled
    else { ga.op[c] = 2; ga.f[c] = f; }
  }
  const unsigned gvb = (unsigned)(((n_video + 7) / 8 + 255) / 256), gab = (unsigned)(((n_audio + 7) / 8 + 255) / 256);
  if (channels == 2) {
    hipLaunchKernelGGL(k_pcm_to_f16<2>, dim3(gvb), dim3(256), 0, stream, sv, gv, reinterpret_cast<half_t*>(d_video), stretch_channel_stride(n_video));
    hipLaunchKernelGGL(k_pcm_to_f16<2>, dim3(gab), dim3(256), 0, stream, sa, ga, reinterpret_cast<half_t*>(d_audio), stretch_channel_stride(n_audio));
  } else {
    hipLaunchKernelGGL(k_pcm_to_f16<1>, dim3(gvb), dim3(256), 0, stream, sv, gv, reinterpret_cast<half_t*>(d_video), stretch_channel_stride(n_video));
    hipLaunchKernelGGL(k_pcm_to_f16<1>, dim3(gab), dim3(256), 0, stream, sa, ga, reinterpret_cast<half_t*>(d_audio), stretch_channel_stride(n_audio));
  }
  S_TRY(hipGetLastError());
  return 0;
}

int stretch_finish(StretchState* s, hipStream_t stream, uint16_t* d_video, int64_t n_video, int channels,
                   int16_t* d_out_interleaved, std::string& err) {
  const int64_t stride = stretch_channel_stride(n_video);
  S_TRY(s->peak.ensure(64));
  S_TRY(hipMemsetAsync(s->peak.p, 0, 4, stream));
  hipLaunchKernelGGL(k_absmax, dim3(2048), dim3(256), 0, stream, reinterpret_cast<half_t*>(d_video), n_video, channels, stride,
                     s->peak.as<unsigned int>());
  const unsigned gb = (unsigned)(((n_video + 7) / 8 + 255) / 256);
  if (channels == 2)
    hipLaunchKernelGGL(k_finish<2>, dim3(gb), dim3(256), 0, stream, reinterpret_cast<half_t*>(d_video), n_video, stride,
                       s->peak.as<unsigned int>(), d_out_interleaved);
  else
    hipLaunchKernelGGL(k_finish<1>, dim3(gb), dim3(256), 0, stream, reinterpret_cast<half_t*>(d_video), n_video, stride,
                       s->peak.as<unsigned int>(), d_out_interleaved);
  S_TRY(hipGetLastError());
  return 0;
}

int stretch_replace(StretchState* s, hipStream_t stream, uint16_t* d_video_u, int64_t n_video, const uint16_t* d_audio_u,
                    int64_t n_audio, int channels, const double* audio_times, const double* video_times, int n_nodes,
                    bool no_pitch_correction, StretchTimes& tm, std::string& err) {
  const int64_t v_stride = stretch_channel_stride(n_video), a_stride = stretch_channel_stride(n_audio);
  half_t* d_video = reinterpret_cast<half_t*>(d_video_u);
  const half_t* d_audio = reinterpret_cast<const half_t*>(d_audio_u);
  s->schedules.clear();
  if (int rc = ensure_hann(s, stream, err)) return rc;
  // ---- interval plan (:387-411)
  std::vector<int64_t> xs(n_nodes), ys(n_nodes);
  for (int i = 0; i < n_nodes; ++i) { xs[i] = (int64_t)(audio_times[i] * kRate); ys[i] = (int64_t)(video_times[i] * kRate); }
  std::vector<Interval> plan;
  for (int i = 0; i + 1 < n_nodes; ++i) {
    const int64_t dx = xs[i + 1] - xs[i], dy = ys[i + 1] - ys[i];
    const double slope = (double)dx / (double)dy;
    Interval iv{0, xs[i], xs[i + 1], ys[i], ys[i + 1]};
    if (dy < 2 * kRate || std::fabs(1 - slope) > 0.1) iv.kind = 0;
    else if (no_pitch_correction || std::fabs(1 - slope) <= 0.005 || std::llabs(dy - dx) < kMinOffset) iv.kind = 1;
    else iv.kind = 2;
    if (iv.kind != 0) {
      if (iv.y0 < 0 || iv.y1 > n_video) return sfail(err, DA_ERR_ARG, "replace: video interval [%lld, %lld) outside the %lld video samples", (long long)iv.y0, (long long)iv.y1, (long long)n_video);
      if (iv.kind == 2 && (iv.x0 < 0 || iv.x1 > n_audio)) return sfail(err, DA_ERR_ARG, "replace: audio interval [%lld, %lld) outside the %lld audio samples", (long long)iv.x0, (long long)iv.x1, (long long)n_audio);
    }
    plan.push_back(iv);
  }

  // ---- resampled intervals (:233-244, :412-414)
  {
    std::vector<ResChunk> chunks;
    int64_t coef_total = 0;
    double points = 0;
    for (const Interval& iv : plan) {
      if (iv.kind != 1) continue;
      const int64_t num = iv.y1 - iv.y0;
      const double start = (double)iv.x0, step = ((double)iv.x1 - (double)iv.x0) / (double)num;
      for (int64_t first = 0; first < num; first += kResChunk) {
        ResChunk c{};
        c.first = first; c.count = (int32_t)std::min<int64_t>(kResChunk, num - first);
        c.out_abs = iv.y0 + first; c.start = start; c.step = step;
        const double p_first = (double)first * step + start;
        const double p_last = (double)(first + c.count - 1) * step + start;
        const int64_t b0 = std::max<int64_t>((int64_t)(p_first - 2), 0);
        const int64_t b1 = std::min<int64_t>((int64_t)(p_last + 2), n_audio);
        if (b1 - b0 < 3) return sfail(err, DA_ERR_ARG, "replace: resampling interval reads outside the audio (samples %lld..%lld of %lld)", (long long)b0, (long long)b1, (long long)n_audio);
        c.b0 = b0; c.n = (int32_t)(b1 - b0); c.coef_off = coef_total;
        coef_total += (int64_t)c.n * channels;
        chunks.push_back(c);
      }
      points += (double)num;
    }
    tm.resample_points = points;
    tm.resample_bytes = points * channels * 2.0 /*written*/ + (double)coef_total * 2.0 /*read once*/;
    if (!chunks.empty()) {
      S_TRY(s->res_chunks.ensure(sizeof(ResChunk) * chunks.size()));
      int64_t small_total = 0; bool any_small = false;
      for (ResChunk& c : chunks)
        if (c.n < kFusedMinRows) { c.coef_off = small_total; small_total += (int64_t)c.n * channels; any_small = true; }
      S_TRY(s->coef.ensure(sizeof(double) * (size_t)(small_total + 8)));
      S_TRY(hipMemcpyAsync(s->res_chunks.p, chunks.data(), sizeof(ResChunk) * chunks.size(), hipMemcpyHostToDevice, stream));
      S_TRY(hipEventRecord(s->e0, stream));
      static const SplineRows rows = spline_rows();
      hipLaunchKernelGGL(k_resample_tile, dim3(kMaxTiles, (unsigned)chunks.size(), channels), dim3(256), 0, stream,
                         s->res_chunks.as<ResChunk>(), d_audio, a_stride, d_video, v_stride, rows);
      if (any_small) {
        const int threads = (int)chunks.size() * channels;
        hipLaunchKernelGGL(k_spline_solve, dim3((threads + 63) / 64), dim3(64), 0, stream, s->res_chunks.as<ResChunk>(),
                           (int)chunks.size(), channels, d_audio, a_stride, s->coef.as<double>());
        hipLaunchKernelGGL(k_spline_eval, dim3((kResChunk + 255) / 256, (unsigned)chunks.size()), dim3(256), 0, stream,
                           s->res_chunks.as<ResChunk>(), channels, s->coef.as<double>(), d_video, v_stride);
      }
      S_TRY(hipGetLastError());
      S_TRY(hipEventRecord(s->e1, stream));
      S_TRY(da::stream_wait(stream));
      tm.resample_ms = elapsed(s->e0, s->e1);
    }
  }

  // ---- stretched intervals (:298-385)
  std::vector<const Interval*> st;
  for (const Interval& iv : plan) if (iv.kind == 2) st.push_back(&iv);
  if (st.empty()) return 0;
  const int NS = (int)st.size();
  std::vector<SegDesc> segs(NS);
  std::vector<int32_t> lags_all;
  std::vector<CorrChunk> chunks;
  int64_t table_total = 0, back_total = 0, plan_total = 0, rms_total = 0;
  int max_lags = 0;
  double windows = 0, cbytes = 0;
  for (int k = 0; k < NS; ++k) {
    const Interval& iv = *st[k];
    SegDesc& d = segs[k];
    d.in_off = iv.x0;
    d.n_in = iv.x1 - iv.x0; d.n_out = iv.y1 - iv.y0; d.total = d.n_out - d.n_in; d.n_windows = d.n_in / kSW;
    if (d.n_in < 3 * kSW - 1 || d.n_windows < 2) return sfail(err, DA_ERR_ARG, "Invalid state in Pearson generator.");
    const std::vector<int32_t> l = lag_list(d.total);
    d.n_lags = (int)l.size(); d.lag_off = (int)lags_all.size();
    max_lags = std::max(max_lags, d.n_lags);
    lags_all.insert(lags_all.end(), l.begin(), l.end());
    d.table_off = table_total; table_total += d.n_windows * d.n_lags;
    d.back_off = back_total; back_total += d.n_windows * kND;
    d.plan_off = plan_total; plan_total += d.n_windows + 2;
    corr_chunks(d.n_in, k, rms_total, chunks);
    windows += (double)d.n_windows * d.n_lags;
    cbytes += (double)d.n_in * channels * 2.0 * (1 + d.n_lags);
  }
  const int NC = (int)chunks.size();
  S_TRY(s->segs.ensure(sizeof(SegDesc) * NS));
  S_TRY(s->lags.ensure(sizeof(int32_t) * lags_all.size()));
  S_TRY(s->where.ensure(sizeof(int16_t) * (size_t)table_total));
  S_TRY(s->loss.ensure(sizeof(double) * (size_t)table_total));
  S_TRY(s->back.ensure(sizeof(int16_t) * (size_t)back_total));
  S_TRY(s->plan_in.ensure(sizeof(int64_t) * (size_t)plan_total));
  S_TRY(s->plan_out.ensure(sizeof(int64_t) * (size_t)plan_total));
  S_TRY(s->sched.ensure(sizeof(int64_t) * 2 * (size_t)plan_total));
  S_TRY(s->counts.ensure(sizeof(int32_t) * 2 * NS));
  S_TRY(s->energy.ensure(sizeof(double) * (size_t)rms_total));
  S_TRY(s->corr_chunks.ensure(sizeof(CorrChunk) * NC));
  S_TRY(s->eps.ensure(sizeof(double) * NC));
  S_TRY(hipMemcpyAsync(s->segs.p, segs.data(), sizeof(SegDesc) * NS, hipMemcpyHostToDevice, stream));
  S_TRY(hipMemcpyAsync(s->lags.p, lags_all.data(), sizeof(int32_t) * lags_all.size(), hipMemcpyHostToDevice, stream));
  S_TRY(hipMemcpyAsync(s->corr_chunks.p, chunks.data(), sizeof(CorrChunk) * NC, hipMemcpyHostToDevice, stream));

  // all intervals in one launch each: wavefront per chunk, then wavefront per (chunk, lag)
  S_TRY(hipEventRecord(s->e0, stream));
  hipLaunchKernelGGL(k_fill_table, dim3((unsigned)((table_total + 255) / 256)), dim3(256), 0, stream, s->where.as<int16_t>(),
                     s->loss.as<double>(), table_total);
  hipLaunchKernelGGL(k_chunk_rms, dim3(NC), dim3(64), 0, stream, d_audio, channels, a_stride, s->segs.as<SegDesc>(),
                     s->corr_chunks.as<CorrChunk>(), s->energy.as<double>(), s->eps.as<double>());
  hipLaunchKernelGGL(k_lag_table, dim3(NC, max_lags), dim3(64), 0, stream, d_audio, channels, a_stride,
                     s->segs.as<SegDesc>(), s->corr_chunks.as<CorrChunk>(), s->lags.as<int32_t>(), s->energy.as<double>(),
                     s->eps.as<double>(), s->where.as<int16_t>(), s->loss.as<double>());
  S_TRY(hipGetLastError());
  S_TRY(hipEventRecord(s->e1, stream));
  S_TRY(da::stream_wait(stream));
  tm.correlate_ms = elapsed(s->e0, s->e1); tm.correlate_windows = windows; tm.correlate_bytes = cbytes;

  const size_t lds_bytes = sizeof(double) * (3 * kRow + 2 * kMaxLags) + sizeof(int32_t) * kMaxLags;
  S_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_viterbi), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
  S_TRY(hipEventRecord(s->e0, stream));
  hipLaunchKernelGGL(k_viterbi, dim3(NS), dim3(1024), lds_bytes, stream, s->segs.as<SegDesc>(), s->lags.as<int32_t>(),
                     s->where.as<int16_t>(), s->loss.as<double>(), s->back.as<int16_t>(), s->plan_in.as<int64_t>(),
                     s->plan_out.as<int64_t>(), s->sched.as<int64_t>(), s->counts.as<int32_t>());
  S_TRY(hipGetLastError());
  S_TRY(hipEventRecord(s->e1, stream));
  std::vector<int32_t> counts(2 * NS);
  S_TRY(hipMemcpyAsync(counts.data(), s->counts.p, sizeof(int32_t) * 2 * NS, hipMemcpyDeviceToHost, stream));
  S_TRY(da::stream_wait(stream));
  tm.viterbi_ms = elapsed(s->e0, s->e1);
  for (int k = 0; k < NS; ++k) {
    if (counts[2 * k + 1]) return sfail(err, DA_ERR_STATE, "replace: drift left the +/-%d sample window while back-tracking interval %d", kMaxDrift, k);
    if (counts[2 * k] == 0) return sfail(err, DA_ERR_STATE, "replace: no jump schedule found for interval %d", k);   // the reference fails indexing an empty array (:367)
  }

  S_TRY(hipEventRecord(s->e0, stream));
  double spoints = 0;
  for (int k = 0; k < NS; ++k) {
    const Interval& iv = *st[k];
    const SegDesc& d = segs[k];
    SpliceArgs sa{};
    sa.seg = d_audio + iv.x0; sa.n_in = d.n_in; sa.in_stride = a_stride;
    sa.out = d_video + iv.y0; sa.n_out = d.n_out; sa.out_stride = v_stride; sa.channels = channels;
    sa.pin = s->plan_in.as<int64_t>() + d.plan_off; sa.pout = s->plan_out.as<int64_t>() + d.plan_off;
    sa.count = s->counts.as<int32_t>() + 2 * k;
    sa.rise = s->hann.as<double>(); sa.fall = s->hann.as<double>() + kSW;
    hipLaunchKernelGGL(k_splice, dim3((unsigned)((d.n_out + 255) / 256)), dim3(256), 0, stream, sa);
    spoints += (double)d.n_out;
  }
  S_TRY(hipGetLastError());
  S_TRY(hipEventRecord(s->e1, stream));
  // schedules for the caller / tests
  s->schedules.resize(NS);
  for (int k = 0; k < NS; ++k) {
    s->schedules[k].resize(2 * (size_t)counts[2 * k]);
    if (counts[2 * k])
      S_TRY(hipMemcpyAsync(s->schedules[k].data(), s->sched.as<int64_t>() + 2 * segs[k].plan_off,
                           sizeof(int64_t) * 2 * (size_t)counts[2 * k], hipMemcpyDeviceToHost, stream));
  }
  S_TRY(da::stream_wait(stream));
  tm.splice_ms = elapsed(s->e0, s->e1); tm.splice_points = spoints;
  return 0;
}

}  // namespace da
