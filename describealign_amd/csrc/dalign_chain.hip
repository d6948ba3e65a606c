// Stage-2 heaviest-chain dynamic programme on the GPU (gfx950).
//
// Reference: describealign.py:654-656, :674-697.  Over the verified matches sorted by (audio frame
// i, video frame v) the reference keeps a staircase frontier in a SortedList and gives every match
// the best predecessor with v' <= v; the path is back-tracked from the heaviest entry.  Restated
// (as in the host version, da_chain with a NULL context):
//
//     f[k] = q[k] + max{ f[k'] : k' < k, v[k'] <= v[k] }   (0 when there is none)
//
// with the maximum taken lexicographically over (f, k') -- equal sums resolve to the LATER point,
// which is what the frontier's eviction rule (:679-680) does.  Every f is formed as
// "predecessor's f plus q", one IEEE double addition, so the sums -- and therefore every
// comparison and tie -- are bit-identical to the reference's whatever the evaluation order.
//
// The recurrence is sequential over audio rows.  One persistent wavefront per pair walks the
// rows; the points of a row (sorted by v, typically 10-40) sit one per lane:
//   * prefix maximum over the video ranks <= r from a Fenwick tree whose nodes are 16-byte
//     (sum, id) records: the levels with span >= 2^S live in LDS, the S lowest levels in global
//     memory (L2 resident); the <= S + 16 node addresses of a query depend on r alone, so all
//     loads of a row -- queries AND the nodes the row will update -- are issued before any is used;
//   * points of the same row may chain (v' < v): f[k] = q[k] + max(g[k], f[k-1]).  Since q > 0 the
//     f of a row increase with the lane, so "best earlier point of the row" is the left neighbour:
//     a Jacobi sweep with one DPP wave shift per step, exact in-order double additions;
//   * tree update without atomics: a lane walks its update path only up to the first node that
//     also covers the next lane's rank -- from there on the next lane's larger f wins anyway --
//     so the node sets written by the lanes of one step are disjoint.
// Many pairs run concurrently (one workgroup each, own stream) beside the similarity GEMM of
// later pairs; nothing of the match list ever goes to the host, only the path does.
#include "dalign_common.h"
#include <hipcub/hipcub.hpp>

namespace da {

namespace {

constexpr int kLowMax = 8;        // S <= 8 lowest tree levels in global memory
constexpr int kHighMax = 16;      // up to 16 levels in LDS: ranks < 2^24

__device__ __forceinline__ bool beats(double ac, uint32_t ai, double bc, uint32_t bi) {
  return ac > bc || (ac == bc && ai > bi);
}
__device__ __forceinline__ double node_cum(const uint4& n) { return __hiloint2double((int)n.y, (int)n.x); }
__device__ __forceinline__ uint4 make_node(double c, uint32_t id1) {
  return uint4{(uint32_t)__double2loint(c), (uint32_t)__double2hiint(c), id1, 0u};
}
// value of the lane to the left (lane 0: `edge`), whole-wave shift by one lane
__device__ __forceinline__ double left_neighbour(double x, double edge) {
  const int lo = __builtin_amdgcn_update_dpp(__double2loint(edge), __double2loint(x), 0x138, 0xf, 0xf, false);
  const int hi = __builtin_amdgcn_update_dpp(__double2hiint(edge), __double2hiint(x), 0x138, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double max_f64(double a, double b) {     // plain v_max_f64 (no NaNs here)
  double r;
  asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ double read_lane(double x, int l) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(x), l);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(x), l);
  return __hiloint2double(hi, lo);
}

}  // namespace

// per match: video rank (1-based), row-head flag, input validation
__global__ __launch_bounds__(256) void k_chain_prep(const unsigned long long* __restrict__ keys, const double* __restrict__ q,
                                                    int64_t n, const int32_t* __restrict__ rankmap, int64_t rankmap_len,
                                                    int32_t* __restrict__ rank, uint8_t* __restrict__ flags,
                                                    int32_t* __restrict__ err) {
  const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n) return;
  const unsigned long long key = keys[k];
  const uint32_t i = (uint32_t)(key >> 32), v = (uint32_t)key;
  bool head = true;
  if (k > 0) {
    const unsigned long long prev = keys[k - 1];
    head = (uint32_t)(prev >> 32) != i;
    if (prev >= key) atomicOr(err, 2);                  // not strictly sorted by (i, v)
  }
  flags[k] = head ? 1 : 0;
  if (rankmap) {
    int32_t r = 0;
    if ((int64_t)v < rankmap_len) r = rankmap[v];
    if (r <= 0) atomicOr(err, 4);                       // video frame is not one of the matched rows
    rank[k] = r;
  }
  const double qq = q[k];
  if (!(qq > 0.0) || !(qq < 1e300)) atomicOr(err, 1);   // the reference's qualities are in (0, 50] (:672)
}

__global__ __launch_bounds__(256) void k_rankmap(const int32_t* __restrict__ vlist, int64_t n_v, int32_t* __restrict__ rankmap) {
  const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r < n_v) rankmap[vlist[r]] = (int32_t)r + 1;
}

__global__ __launch_bounds__(64) void k_chain_forward(ChainArgs a) {
  extern __shared__ uint4 s_hi[];                    // tree levels with span >= 2^S: node h covers ranks ((h - lowbit(h)) << S, h << S]
  const int lane = threadIdx.x;
  const uint32_t n_ranks = (uint32_t)a.n_ranks;
  const int S = a.S;
  const uint32_t H = n_ranks >> S;
  for (uint32_t h = lane; h <= H; h += 64) s_hi[h] = uint4{0u, 0u, 0u, 0u};
  __syncthreads();
  uint4* __restrict__ lo = a.tree_lo;                // index = rank; [0] stays empty (the "no node" slot)
  const int32_t n_rows = *a.d_nrows;
  const uint32_t lowmask = (1u << S) - 1u;
  const double NEG = -__builtin_huge_val();
  double best_c = 0.0;
  uint32_t best_i = 0u;                              // id + 1; 0 = none
  const int32_t n = (int32_t)a.n;

  int32_t rb = n_rows > 0 ? a.row_start[0] : 0;
  int32_t re = n_rows > 1 ? a.row_start[1] : n;
  // first chunk of the current row, prefetched one row ahead
  uint32_t r_pf = 0u; double q_pf = 0.0;
  if (n_rows > 0 && rb + lane < re) { r_pf = (uint32_t)a.rank[rb + lane]; q_pf = a.q[rb + lane]; }

  for (int32_t row = 0; row < n_rows; ++row) {
    const int32_t nb = re;
    const int32_t ne = (row + 2 < n_rows) ? a.row_start[row + 2] : n;
    uint32_t r_nx = 0u; double q_nx = 0.0;
    if (row + 1 < n_rows && nb + lane < ne) { r_nx = (uint32_t)a.rank[nb + lane]; q_nx = a.q[nb + lane]; }

    double carry = NEG;
    for (int32_t c = rb; c < re; c += 64) {
      const int cnt = (re - c) < 64 ? (re - c) : 64;
      const int32_t k = c + lane;
      const bool valid = lane < cnt;
      uint32_t r; double qv;
      if (c == rb) { r = valid ? r_pf : 0u; qv = valid ? q_pf : 0.0; }
      else { r = valid ? (uint32_t)a.rank[k] : 0u; qv = valid ? a.q[k] : 0.0; }
      uint32_t rnext = (uint32_t)__shfl_down((int)r, 1);
      if (lane + 1 >= cnt) rnext = 0xFFFFFFFFu;

      // ---- all tree loads of this step: query nodes and the nodes of the update path
      uint4 ql[kLowMax], qh[kHighMax], ul[kLowMax], uh[kHighMax];
      uint32_t uli[kLowMax], uhi[kHighMax];
      {
        uint32_t x = r;
#pragma unroll
        for (int t = 0; t < kLowMax; ++t) {
          const bool on = (x & lowmask) != 0u;
          ql[t] = uint4{0u, 0u, 0u, 0u};
          if (__any(on)) {
            ql[t] = lo[on ? x : 0u];
            x = on ? (x & (x - 1u)) : x;
          }
        }
        uint32_t xh = r >> S;
#pragma unroll
        for (int t = 0; t < kHighMax; ++t) {
          const bool on = xh != 0u;
          qh[t] = uint4{0u, 0u, 0u, 0u};
          if (__any(on)) {
            qh[t] = s_hi[on ? xh : 0u];
            xh = on ? (xh & (xh - 1u)) : xh;
          }
        }
        uint32_t ux = r;
        bool live = valid;
#pragma unroll
        for (int t = 0; t < kLowMax; ++t) {
          live = live && ux <= n_ranks && ux < rnext;
          const bool on = live && (ux & lowmask) != 0u;
          uli[t] = on ? ux : 0u;
          ul[t] = uint4{0u, 0u, 0u, 0u};
          if (__any(on)) {
            ul[t] = lo[uli[t]];
            ux = on ? ux + (ux & (0u - ux)) : ux;
          }
        }
#pragma unroll
        for (int t = 0; t < kHighMax; ++t) {
          live = live && ux <= n_ranks && ux < rnext;
          const bool on = live;                        // ux is a multiple of 2^S here
          uhi[t] = on ? (ux >> S) : 0u;
          uh[t] = uint4{0u, 0u, 0u, 0u};
          if (__any(on)) {
            uh[t] = s_hi[uhi[t]];
            ux = on ? ux + (ux & (0u - ux)) : ux;
          }
        }
      }
      // ---- best predecessor among earlier rows (and earlier chunks of this row)
      double gc = 0.0; uint32_t gi = 0u;
#pragma unroll
      for (int t = 0; t < kLowMax; ++t) {
        const double cc = node_cum(ql[t]);
        if (beats(cc, ql[t].z, gc, gi)) { gc = cc; gi = ql[t].z; }
      }
#pragma unroll
      for (int t = 0; t < kHighMax; ++t) {
        const double cc = node_cum(qh[t]);
        if (beats(cc, qh[t].z, gc, gi)) { gc = cc; gi = qh[t].z; }
      }
      // ---- chaining inside the row: f[s] = q[s] + max(g[s], f[s-1]), exact and in order
      double f = NEG;
      for (int it = 0; it < cnt; ++it) f = qv + max_f64(gc, left_neighbour(f, carry));
      const double fp = left_neighbour(f, carry);
      const bool from_row = fp >= gc;                  // the row's own point is the later one: it wins ties
      const uint32_t id1 = (uint32_t)k + 1u;
      if (valid) a.pred[k] = from_row ? (k - 1) : ((int32_t)gi - 1);
      // ---- tree update (disjoint node sets per lane, see header)
#pragma unroll
      for (int t = 0; t < kLowMax; ++t)
        if (uli[t] != 0u && beats(f, id1, node_cum(ul[t]), ul[t].z)) lo[uli[t]] = make_node(f, id1);
#pragma unroll
      for (int t = 0; t < kHighMax; ++t)
        if (uhi[t] != 0u && beats(f, id1, node_cum(uh[t]), uh[t].z)) s_hi[uhi[t]] = make_node(f, id1);
      // the row's last point carries its largest sum
      const double fl = read_lane(f, cnt - 1);
      carry = fl;
      if (fl >= best_c) { best_c = fl; best_i = (uint32_t)(c + cnt); }
    }
    rb = nb; re = ne; r_pf = r_nx; q_pf = q_nx;
  }
  if (lane == 0) { a.meta[0] = (int64_t)best_i - 1; a.meta[1] = 0; }
}

// Back-track from the heaviest point through pred[] (:690-697).  The chain's ids decrease, and a
// predecessor is rarely more than a few rows back, so pred[] is pulled through LDS in windows of
// kBackWindow ids and chased there by one lane; ids are staged and written out coalesced.
constexpr int kBackWindow = 12288;
constexpr int kBackStage = 2048;

__global__ __launch_bounds__(256) void k_chain_backtrack(const int32_t* __restrict__ pred, int64_t n, int32_t* __restrict__ path_ids,
                                                         int64_t* __restrict__ meta) {
  __shared__ int32_t s_pred[kBackWindow];
  __shared__ int32_t s_out[kBackStage];
  __shared__ int32_t s_cur, s_nout;
  int64_t total = 0;
  int32_t cur = (int32_t)meta[0];
  while (cur >= 0) {
    const int32_t whi = cur + 1;
    const int32_t wlo = whi > kBackWindow ? whi - kBackWindow : 0;
    for (int32_t t = wlo + (int32_t)threadIdx.x; t < whi; t += 256) s_pred[t - wlo] = pred[t];
    __syncthreads();
    // chase inside the window, flushing the stage when it fills
    while (true) {
      if (threadIdx.x == 0) {
        int32_t m = 0, p = cur;
        while (p >= wlo && m < kBackStage) { s_out[m++] = p; p = s_pred[p - wlo]; }
        s_cur = p; s_nout = m;
      }
      __syncthreads();
      const int32_t m = s_nout;
      cur = s_cur;
      for (int32_t t = threadIdx.x; t < m; t += 256) path_ids[total + t] = s_out[t];
      total += m;
      __syncthreads();
      if (cur < wlo || m < kBackStage) break;
    }
    // cur < wlo here (or the chain ended); pred ids are < their own id, so the next window starts at cur
  }
  if (threadIdx.x == 0) meta[1] = total;
}

// ascending (audio frame, video frame) arrays from the descending id list
__global__ __launch_bounds__(256) void k_chain_gather(const unsigned long long* __restrict__ keys, const int32_t* __restrict__ path_ids,
                                                      const int64_t* __restrict__ meta, int32_t* __restrict__ out_i,
                                                      int32_t* __restrict__ out_v) {
  const int64_t L = meta[1];
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < L; t += (int64_t)gridDim.x * blockDim.x) {
    const unsigned long long key = keys[path_ids[L - 1 - t]];
    out_i[t] = (int32_t)(key >> 32);
    out_v[t] = (int32_t)(uint32_t)key;
  }
}

int chain_tree_shift(int64_t n_ranks) {
  // LDS holds the levels with span >= 2^S: (n_ranks >> S) + 1 nodes of 16 B within 128 KiB
  int S = 6;
  while (S < kLowMax && ((n_ranks >> S) + 2) * 16 > 128 * 1024) ++S;
  return S;
}

size_t chain_rows_temp_bytes(int64_t n) {
  size_t bytes = 0;
  hipcub::CountingInputIterator<int32_t> ids(0);
  (void)hipcub::DeviceSelect::Flagged(nullptr, bytes, ids, (const uint8_t*)nullptr, (int32_t*)nullptr, (int32_t*)nullptr, (int)n);
  return bytes;
}

// per-match ranks, row-head flags, validation (reads the rank map: runs where that is built)
int launch_chain_prep(const ChainLaunch& c, hipStream_t s) {
  if (c.n > 0x7fffffffLL || c.n_ranks >= (1LL << 24)) return -1;
  const int S = chain_tree_shift(c.n_ranks);
  if (((c.n_ranks >> S) + 2) * 16 > 150 * 1024) return -1;
  if (c.n <= 0) return 0;
  const unsigned blocks = (unsigned)((c.n + 255) / 256);
  hipLaunchKernelGGL(k_chain_prep, dim3(blocks), dim3(256), 0, s, c.keys, c.q, c.n, c.rankmap, c.rankmap_len, c.rank, c.flags, c.err);
  return 0;
}

// row starts, the forward DP, the back-track and the gather, all on stream s
int launch_chain_dp(const ChainLaunch& c, hipStream_t s) {
  if (c.n <= 0) return 0;
  const int S = chain_tree_shift(c.n_ranks);
  hipcub::CountingInputIterator<int32_t> ids(0);
  size_t bytes = c.temp_bytes;
  if (hipcub::DeviceSelect::Flagged(c.temp, bytes, ids, c.flags, c.row_start, c.d_nrows, (int)c.n, s) != hipSuccess) return -1;
  ChainArgs a{};
  a.q = c.q; a.rank = c.rank; a.row_start = c.row_start; a.d_nrows = c.d_nrows; a.n = c.n;
  a.n_ranks = c.n_ranks; a.S = S; a.tree_lo = reinterpret_cast<uint4*>(c.tree_lo); a.pred = c.pred; a.meta = c.meta;
  const size_t lds = (size_t)((c.n_ranks >> S) + 2) * 16;
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_chain_forward), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL(k_chain_forward, dim3(1), dim3(64), lds, s, a);
  hipLaunchKernelGGL(k_chain_backtrack, dim3(1), dim3(256), 0, s, c.pred, c.n, c.path_ids, c.meta);
  hipLaunchKernelGGL(k_chain_gather, dim3(256), dim3(256), 0, s, c.keys, c.path_ids, c.meta, c.out_i, c.out_v);
  return 0;
}

void launch_rankmap(const int32_t* vlist, int64_t n_v, int32_t* rankmap, hipStream_t s) {
  if (n_v <= 0) return;
  hipLaunchKernelGGL(k_rankmap, dim3((unsigned)((n_v + 255) / 256)), dim3(256), 0, s, vlist, n_v, rankmap);
}

}  // namespace da
