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
// The recurrence is sequential over audio rows.  Two implementations:
//   * k_chain_columns (the default, further down): the video ranks cut into columns, one wavefront per
//     column with its Fenwick tree in LDS, the columns a pipeline over the rows -- the whole chip works
//     on one pair's DP (2 h pair: 72 ms);
//   * k_chain_forward / k_chain_forward_w4 (rounds 1-2, DALIGN_CHAIN_KERNEL=rows): ONE persistent
//     workgroup per pair walks the rows, the points of a row one per lane, prefix maxima from a Fenwick
//     tree split between LDS (levels with span >= 2^S) and global memory (the S lowest levels, L2
//     resident), in-row chaining by Jacobi sweeps over a DPP wave shift, tree updates without atomics
//     (a lane walks its update path only up to the first node that also covers the next lane's rank).
//     1.75 s per 2 h pair; kept as an independent cross-check of the column pipeline in the GPU tests.
// Nothing of the match list ever goes to the host, only the path does.
#include "dalign_common.h"
#include <hipcub/hipcub.hpp>
#include <algorithm>
#include <cstdlib>

namespace da {

namespace {

constexpr int kScrap = 256;        // scrap records behind either part of the tree, one per thread (up to 4 wavefronts)
constexpr int kLowMax = 8;        // S <= 8 lowest tree levels in global memory; up to 16 levels in LDS: ranks < 2^24

__device__ __forceinline__ bool beats(double ac, uint32_t ai, double bc, uint32_t bi) {
  return ac > bc || (ac == bc && ai > bi);
}
__device__ __forceinline__ double node_cum(const uint4& n) { return __hiloint2double((int)n.y, (int)n.x); }
__device__ __forceinline__ uint4 make_node(double c, uint32_t id1) {
  return uint4{(uint32_t)__double2loint(c), (uint32_t)__double2hiint(c), id1, 0u};
}
// value of the lane to the left (lane 0 reads 0.0), whole-wave shift by one lane
__device__ __forceinline__ double left_neighbour(double x) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(x), 0x138, 0xf, 0xf, true);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(x), 0x138, 0xf, 0xf, true);
  return __hiloint2double(hi, lo);
}
// one v_max_f64: this file is compiled with -ffinite-math-only (no NaN canonicalisation, and no
// hazard padding as around inline assembly); nothing here is ever NaN or infinite
__device__ __forceinline__ double max_f64(double a, double b) { return __builtin_fmax(a, b); }
__device__ __forceinline__ double read_lane(double x, int l) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(x), l);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(x), l);
  return __hiloint2double(hi, lo);
}

}  // namespace

// per match: video rank (1-based), row-head flag, input validation
__global__ __launch_bounds__(256) void k_chain_prep(const unsigned long long* __restrict__ keys, const double* __restrict__ q,
                                                    int64_t n, const int32_t* __restrict__ rankmap, int64_t rankmap_len,
                                                    const int32_t* __restrict__ dense, int32_t* __restrict__ rank,
                                                    uint8_t* __restrict__ flags, int32_t* __restrict__ err) {
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
    if (r <= 0) { atomicOr(err, 4); r = 1; }            // video frame is not one of the matched rows: reported by the host; rank 1 keeps every later kernel inside its arrays
    else if (dense) r = dense[v] + 1;                   // rank among the frames that HAVE a match (launch_dense_ranks): the columns' trees cover only those
    rank[k] = r;
  }
  // the reference's qualities are in (0, 50] (:672).  Tested on the bit pattern: this file is compiled with
  // -ffinite-math-only, under which a floating-point comparison need not reject a NaN -- and a NaN pattern would win every
  // ds_max_u64 of the column kernel.  0 < q < 1e300  <=>  0 < bits < bits(1e300), read as a signed integer.
  const long long qbits = __double_as_longlong(q[k]);
  if (!(qbits > 0 && qbits < 0x7E37E43C8800759CLL)) atomicOr(err, 1);
}

__global__ __launch_bounds__(256) void k_rankmap(const int32_t* __restrict__ vlist, int64_t n_v, int32_t* __restrict__ rankmap) {
  const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r < n_v) rankmap[vlist[r]] = (int32_t)r + 1;
}

// LOW = S = tree levels in global memory, HIGH >= number of levels in LDS: every step loads exactly
// LOW + HIGH query nodes and LOW + HIGH update-path nodes per lane, unconditionally (slot 0 of
// either array is the permanently empty "no node" record), so that one s_waitcnt covers them all --
// a conditional load per level costs a full memory round trip each.
#ifdef DA_CHAIN_STAMPS          // diagnostic build only: cycles per phase into meta[2..7]
#define DA_STAMP(k) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); stamp_acc[k] += t_ - stamp_t; stamp_t = t_; }
#else
#define DA_STAMP(k)
#endif

// Placement.  The DP is ONE workgroup that runs for seconds beside the similarity GEMMs of later pairs.
// The hardware deals the workgroups of a grid round-robin over the 8 XCDs, so a one-block grid would
// put every pair's DP on the same XCD -- and a GEMM, whose workgroups are dealt statically over the
// XCDs too, then waits for the one XCD that has lost a quarter of its CUs.  The kernel is therefore
// launched with one block per XCD; the block that finds itself on XCD `a.xcd` does the work and the
// others leave at once.  Nothing but speed depends on the placement: if no block sits on the wanted
// XCD, the block that arrives last takes the job.
__device__ __forceinline__ bool chain_block_elected(const ChainArgs& a) {
  __shared__ int s_me;
  if (threadIdx.x == 0) {
    const unsigned xcc = __builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)) & 0xfu;      // HW_REG_XCC_ID[3:0]
    int me = 0;
    if (gridDim.x == 1) me = 1;
    else {
      if ((int)xcc == a.xcd) me = atomicCAS(a.claim, 0, 1) == 0;
      const int arrived = atomicAdd(a.claim + 1, 1);
      if (!me && arrived == (int)gridDim.x - 1) me = atomicCAS(a.claim, 0, 1) == 0;
    }
    s_me = me;
  }
  __syncthreads();
  return s_me != 0;
}

template <int LOW, int HIGH>
__global__ __launch_bounds__(64) void k_chain_forward(ChainArgs a) {
  if (!chain_block_elected(a)) return;
#ifdef DA_CHAIN_STAMPS
  unsigned long long stamp_acc[6] = {0, 0, 0, 0, 0, 0};
  unsigned long long stamp_t = __builtin_amdgcn_s_memtime();
#endif
  extern __shared__ uint4 s_hi[];                    // tree levels with span >= 2^S: node h covers ranks ((h - lowbit(h)) << S, h << S]
  const int lane = threadIdx.x;
  const uint32_t n_ranks = (uint32_t)a.n_ranks;
  constexpr int S = LOW;
  const uint32_t H = n_ranks >> S;
  for (uint32_t h = lane; h <= H; h += 64) s_hi[h] = uint4{0u, 0u, 0u, 0u};
  __syncthreads();
  uint4* __restrict__ lo = a.tree_lo;                // index = rank; [0] stays empty
  const int32_t n_rows = *a.d_nrows;
  constexpr uint32_t lowmask = (1u << S) - 1u;
  const double NEG = -1.0e300;                       // below every sum; adding a quality leaves it there
  double best_c = 0.0;
  uint32_t best_i = 0u;                              // id + 1; 0 = none
  uint32_t best_r = 0xFFFFFFFFu;                     // its video rank
  const int32_t n = (int32_t)a.n;

  // Row boundaries: 64 at a time, one per lane, two blocks held (the current one and the next), so the
  // scalar "where does the next row end" never waits for memory.  rs(j) = first match of row j (n past the end).
  int32_t rs_base = 0;
  auto load_block = [&](int32_t base) -> int32_t {
    const int32_t j = base + lane;
    return j < n_rows ? a.row_start[j] : n;
  };
  int32_t rs_cur = load_block(0), rs_nxt = load_block(64);
  auto rs = [&](int32_t j) -> int32_t {               // j in [rs_base, rs_base + 128)
    const int32_t o = j - rs_base;
    return o < 64 ? __builtin_amdgcn_readlane(rs_cur, o) : __builtin_amdgcn_readlane(rs_nxt, o - 64);
  };
  int32_t row = 0;
  int32_t rb = n_rows > 0 ? rs(0) : 0, re = n_rows > 0 ? rs(1) : 0;
  int32_t c = rb;
  // the step's points (video rank, quality), prefetched one step ahead
  uint32_t r_cur = 0u; double q_cur = 0.0;
  if (c + lane < re) { r_cur = (uint32_t)a.rank[c + lane]; q_cur = a.q[c + lane]; }
  double carry = NEG;

  while (row < n_rows) {
    {
      const int cnt = (re - c) < 64 ? (re - c) : 64;
      const int32_t k = c + lane;
      const bool valid = lane < cnt;
      const uint32_t r = valid ? r_cur : 0u;
      const double qv = valid ? q_cur : 0.0;
      uint32_t rnext = (uint32_t)__shfl_down((int)r, 1);
      if (lane + 1 >= cnt) rnext = 0xFFFFFFFFu;
      // where the next step is: the rest of this row, or the start of the next row
      const bool same_row = c + 64 < re;
      if (!same_row && row + 1 - rs_base >= 64) {      // slide the boundary blocks
        rs_base += 64; rs_cur = rs_nxt; rs_nxt = load_block(rs_base + 64);
      }
      const int32_t n_c = same_row ? c + 64 : re;
      const int32_t n_re = same_row ? re : (row + 1 < n_rows ? rs(row + 2) : n);

      DA_STAMP(0)
      // ---- all tree loads of this step: query nodes and the nodes of the update path.
      // Shortcut: the heaviest point so far is the prefix maximum of every rank at or right of its
      // own, so a step whose points all lie there (about half of the steps of a wide row) needs no
      // query at all.
      const bool no_query = __all(!valid || r >= best_r);
      uint4 ql[LOW], qh[HIGH], ul[LOW], uh[HIGH];
      uint32_t uli[LOW], uhi[HIGH];
      // Query path, level by level: the node of span 2^t on it is r with the bits below t cleared,
      // present iff bit t of r is set (no dependence between levels: plain bit masks).
      if (!no_query) {
#pragma unroll
        for (int t = 0; t < LOW; ++t)
          ql[t] = lo[(r & (1u << t)) ? (r & ~((1u << t) - 1u)) : 0u];
        const uint32_t rh = r >> S;
#pragma unroll
        for (int t = 0; t < HIGH; ++t)
          qh[t] = s_hi[(rh & (1u << t)) ? (rh & ~((1u << t) - 1u)) : 0u];     // 0: the empty record
      }
      // Update path, level by level: r rounded up to a multiple of 2^t is the node when its quotient
      // is odd (otherwise it belongs to a higher level); the walk stops below `limit`: past the last
      // rank, or at the first node that also covers the next lane's rank.
      {
        const uint32_t limit = valid ? (rnext <= n_ranks ? rnext : n_ranks + 1u) : 0u;
#pragma unroll
        for (int t = 0; t < LOW; ++t) {
          const uint32_t y = (r + ((1u << t) - 1u)) & ~((1u << t) - 1u);
          uli[t] = ((y & (1u << t)) && y < limit) ? y : 0u;
          ul[t] = lo[uli[t]];
        }
        const uint32_t rh = (r + lowmask) >> S;        // r rounded up to a multiple of 2^S, in units of 2^S
        const uint32_t limit_h = (limit + lowmask) >> S;                       // y << S < limit  <=>  y < ceil(limit / 2^S)
#pragma unroll
        for (int t = 0; t < HIGH; ++t) {
          const uint32_t y = (rh + ((1u << t) - 1u)) & ~((1u << t) - 1u);
          uhi[t] = ((y & (1u << t)) && y < limit_h) ? y : 0u;
          uh[t] = s_hi[uhi[t]];
        }
      }
      // next step's points: issued behind the tree loads, so the (in-order) wait for those does not
      // include this HBM round trip
      uint32_t r_nx = 0u; double q_nx = 0.0;
      if (n_c + lane < n_re) { r_nx = (uint32_t)a.rank[n_c + lane]; q_nx = a.q[n_c + lane]; }
      DA_STAMP(1)
      // ---- best predecessor among earlier rows (and earlier chunks of this row): the largest sum,
      // and among equal sums the largest id
      double gc = best_c; uint32_t gi = best_i;
      if (!no_query) {
        gc = 0.0; gi = 0u;
#pragma unroll
        for (int t = 0; t < LOW; ++t) gc = max_f64(gc, node_cum(ql[t]));
#pragma unroll
        for (int t = 0; t < HIGH; ++t) gc = max_f64(gc, node_cum(qh[t]));
#pragma unroll
        for (int t = 0; t < LOW; ++t) gi = (node_cum(ql[t]) == gc && ql[t].z > gi) ? ql[t].z : gi;
#pragma unroll
        for (int t = 0; t < HIGH; ++t) gi = (node_cum(qh[t]) == gc && qh[t].z > gi) ? qh[t].z : gi;
      }
      DA_STAMP(2)
      // ---- chaining inside the row: f[s] = q[s] + max(g[s], f[s-1]), exact and in order.  A Jacobi
      // sweep: every lane re-evaluates from its left neighbour; the values only grow, so a block of
      // sweeps that changes nothing is the fixed point (left of the heaviest point runs are short).
      // Lane 0's left neighbour is the last point of the row's previous step (`carry`); folded into
      // its g, so the shift can feed it a zero (sums are >= 0).
      const double ge = lane == 0 ? max_f64(gc, carry) : gc;
      double f = qv + ge;                              // sweep 0
      for (int it = 1; it < cnt; it += 8) {
        const double f0 = f;
#pragma unroll
        for (int u = 0; u < 8; ++u) f = qv + max_f64(ge, left_neighbour(f));
        if (!__any(valid && f != f0)) break;
      }
      DA_STAMP(3)
      const double fleft = left_neighbour(f);          // by ALL lanes: a DPP read of a lane masked off by a branch returns 0
      const double fp = lane == 0 ? carry : fleft;
      const bool from_row = fp >= gc;                  // the row's own point is the later one: it wins ties
      const uint32_t id1 = (uint32_t)k + 1u;
      if (valid) a.pred[k] = from_row ? (k - 1) : ((int32_t)gi - 1);
      // ---- tree update (disjoint node sets per lane, see header).  This point is the latest, so
      // it also wins every tie: f >= node's sum.
      // Branch-free: a lane with nothing to write stores to its own scrap record behind the tree (one
      // shared scrap address would serialise the 64 lanes of an LDS write).
      {
        const uint4 me = make_node(f, id1);
#pragma unroll
        for (int t = 0; t < LOW; ++t)
          lo[(uli[t] != 0u && f >= node_cum(ul[t])) ? uli[t] : n_ranks + 1u + lane] = me;
#pragma unroll
        for (int t = 0; t < HIGH; ++t)
          s_hi[(uhi[t] != 0u && f >= node_cum(uh[t])) ? uhi[t] : H + 1u + lane] = me;
      }
      DA_STAMP(4)
      // the row's last point carries its largest sum
      const double fl = read_lane(f, cnt - 1);
      if (fl >= best_c) { best_c = fl; best_i = (uint32_t)(c + cnt); best_r = (uint32_t)__builtin_amdgcn_readlane((int)r, cnt - 1); }
      carry = same_row ? fl : NEG;
      if (!same_row) ++row;
      c = n_c; re = n_re; r_cur = r_nx; q_cur = q_nx;
    }
  }
  if (lane == 0) { a.meta[0] = (int64_t)best_i - 1; a.meta[1] = 0; }
#ifdef DA_CHAIN_STAMPS
  if (lane == 0) for (int t = 0; t < 6; ++t) a.meta[2 + t] = (int64_t)stamp_acc[t];
#endif
}

// Four-wavefront form for wide rows (a 2 h pair has ~255 matches per audio row): a "super-step" takes up
// to 256 matches of one row, 64 per wavefront.  Tree loads, reductions and tree updates of the four
// chunks run in parallel on the CU's four SIMDs (every query of a row sees the tree as it was when the
// row began; what the row's own earlier matches contribute is exactly the left neighbour in the Jacobi
// chain, so only that chain is sequential: the wavefronts take turns, handing the running sum on
// through LDS).  The node sets written by the lanes of a super-step are disjoint across wavefronts too:
// the walk-stop rule uses the rank of the next match of the row, whichever wavefront holds it.
constexpr int kW4 = 4;
template <int LOW, int HIGH>
__global__ __launch_bounds__(64 * kW4) void k_chain_forward_w4(ChainArgs a) {
  if (!chain_block_elected(a)) return;
  extern __shared__ uint4 s_hi[];
  __shared__ double s_carry, s_best_c;
  __shared__ uint32_t s_best_i, s_best_r;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const uint32_t n_ranks = (uint32_t)a.n_ranks;
  constexpr int S = LOW;
  const uint32_t H = n_ranks >> S;
  for (uint32_t h = tid; h <= H; h += 64 * kW4) s_hi[h] = uint4{0u, 0u, 0u, 0u};
  const double NEG = -1.0e300;
  if (tid == 0) { s_carry = NEG; s_best_c = 0.0; s_best_i = 0u; s_best_r = 0xFFFFFFFFu; }
  __syncthreads();
  uint4* __restrict__ lo = a.tree_lo;
  const int32_t n_rows = *a.d_nrows;
  constexpr uint32_t lowmask = (1u << S) - 1u;
  const int32_t n = (int32_t)a.n;

  int32_t rs_base = 0;
  auto load_block = [&](int32_t base) -> int32_t {
    const int32_t j = base + lane;
    return j < n_rows ? a.row_start[j] : n;
  };
  int32_t rs_cur = load_block(0), rs_nxt = load_block(64);
  auto rs = [&](int32_t j) -> int32_t {
    const int32_t o = j - rs_base;
    return o < 64 ? __builtin_amdgcn_readlane(rs_cur, o) : __builtin_amdgcn_readlane(rs_nxt, o - 64);
  };
  int32_t row = 0;
  int32_t rb = n_rows > 0 ? rs(0) : 0, re = n_rows > 0 ? rs(1) : 0;
  int32_t c = rb;
  // this thread's match of the super-step (video rank, quality, rank of the row's next match), prefetched
  auto fetch = [&](int32_t base, int32_t end, uint32_t& r, double& q, uint32_t& rn) {
    const int32_t k = base + tid;
    r = 0u; q = 0.0; rn = 0xFFFFFFFFu;
    if (k < end) { r = (uint32_t)a.rank[k]; q = a.q[k]; if (k + 1 < end) rn = (uint32_t)a.rank[k + 1]; }
  };
  uint32_t r_cur, rn_cur; double q_cur;
  fetch(c, re, r_cur, q_cur, rn_cur);

  while (row < n_rows) {
    const int cnt_all = (re - c) < 64 * kW4 ? (re - c) : 64 * kW4;
    const int nw = (cnt_all + 63) >> 6;                 // wavefronts with matches in this super-step
    const int cnt = cnt_all - 64 * wave < 0 ? 0 : (cnt_all - 64 * wave > 64 ? 64 : cnt_all - 64 * wave);
    const int32_t k = c + tid;
    const bool valid = lane < cnt;
    const uint32_t r = valid ? r_cur : 0u;
    const double qv = valid ? q_cur : 0.0;
    const uint32_t rnext = valid ? rn_cur : 0xFFFFFFFFu;
    const double best_c = s_best_c; const uint32_t best_i = s_best_i, best_r = s_best_r;     // as the last super-step left them
    const bool same_row = c + 64 * kW4 < re;
    if (!same_row && row + 1 - rs_base >= 64) { rs_base += 64; rs_cur = rs_nxt; rs_nxt = load_block(rs_base + 64); }
    const int32_t n_c = same_row ? c + 64 * kW4 : re;
    const int32_t n_re = same_row ? re : (row + 1 < n_rows ? rs(row + 2) : n);

    const bool no_query = __all(!valid || r >= best_r);
    uint4 ql[LOW], qh[HIGH], ul[LOW], uh[HIGH];
    uint32_t uli[LOW], uhi[HIGH];
    if (!no_query) {
#pragma unroll
      for (int t = 0; t < LOW; ++t) ql[t] = lo[(r & (1u << t)) ? (r & ~((1u << t) - 1u)) : 0u];
      const uint32_t rh = r >> S;
#pragma unroll
      for (int t = 0; t < HIGH; ++t) qh[t] = s_hi[(rh & (1u << t)) ? (rh & ~((1u << t) - 1u)) : 0u];
    }
    {
      const uint32_t limit = valid ? (rnext <= n_ranks ? rnext : n_ranks + 1u) : 0u;
#pragma unroll
      for (int t = 0; t < LOW; ++t) {
        const uint32_t y = (r + ((1u << t) - 1u)) & ~((1u << t) - 1u);
        uli[t] = ((y & (1u << t)) && y < limit) ? y : 0u;
        ul[t] = lo[uli[t]];
      }
      const uint32_t rh = (r + lowmask) >> S;
      const uint32_t limit_h = (limit + lowmask) >> S;
#pragma unroll
      for (int t = 0; t < HIGH; ++t) {
        const uint32_t y = (rh + ((1u << t) - 1u)) & ~((1u << t) - 1u);
        uhi[t] = ((y & (1u << t)) && y < limit_h) ? y : 0u;
        uh[t] = s_hi[uhi[t]];
      }
    }
    uint32_t r_nx, rn_nx; double q_nx;
    fetch(n_c, n_re, r_nx, q_nx, rn_nx);
    double gc = best_c; uint32_t gi = best_i;
    if (!no_query) {
      gc = 0.0; gi = 0u;
#pragma unroll
      for (int t = 0; t < LOW; ++t) gc = max_f64(gc, node_cum(ql[t]));
#pragma unroll
      for (int t = 0; t < HIGH; ++t) gc = max_f64(gc, node_cum(qh[t]));
#pragma unroll
      for (int t = 0; t < LOW; ++t) gi = (node_cum(ql[t]) == gc && ql[t].z > gi) ? ql[t].z : gi;
#pragma unroll
      for (int t = 0; t < HIGH; ++t) gi = (node_cum(qh[t]) == gc && qh[t].z > gi) ? qh[t].z : gi;
    }
    // ---- the row's own chain: the wavefronts take turns, the running sum travels through LDS
    double f = NEG, carry = NEG;
    for (int t = 0; t < nw; ++t) {
      if (wave == t) {
        carry = s_carry;
        const double ge = lane == 0 ? max_f64(gc, carry) : gc;
        f = qv + ge;
        for (int it = 1; it < cnt; it += 8) {
          const double f0 = f;
#pragma unroll
          for (int u = 0; u < 8; ++u) f = qv + max_f64(ge, left_neighbour(f));
          if (!__any(valid && f != f0)) break;
        }
        const double fl = read_lane(f, cnt - 1);
        if (lane == 0) s_carry = fl;
      }
      __syncthreads();
    }
    const double fleft = left_neighbour(f);
    const double fp = lane == 0 ? carry : fleft;
    const bool from_row = fp >= gc;
    const uint32_t id1 = (uint32_t)k + 1u;
    if (valid) a.pred[k] = from_row ? (k - 1) : ((int32_t)gi - 1);
    {
      const uint4 me = make_node(f, id1);
#pragma unroll
      for (int t = 0; t < LOW; ++t)
        lo[(uli[t] != 0u && f >= node_cum(ul[t])) ? uli[t] : n_ranks + 1u + tid] = me;
#pragma unroll
      for (int t = 0; t < HIGH; ++t)
        s_hi[(uhi[t] != 0u && f >= node_cum(uh[t])) ? uhi[t] : H + 1u + tid] = me;
    }
    if (wave == nw - 1 && lane == cnt - 1) {            // the super-step's last match carries the row's largest sum so far
      if (f >= best_c) { s_best_c = f; s_best_i = id1; s_best_r = r; }
      if (!same_row) s_carry = NEG;                      // next super-step starts a new row
    }
    __syncthreads();                                     // tree, best and carry are published
    if (!same_row) ++row;
    c = n_c; re = n_re; r_cur = r_nx; q_cur = q_nx; rn_cur = rn_nx;
  }
  if (tid == 0) { a.meta[0] = (int64_t)s_best_i - 1; a.meta[1] = 0; }
}

// Back-track from the heaviest point through pred[] (:690-697), in three steps (round 4; rounds 2-3: ONE workgroup that
// pulled all of pred[] through LDS and chased there -- 17 ms per 2 h pair, 26 % of the DP, bound by that workgroup's
// 17 GB/s).  The ids of a chain decrease, so pred[] is cut into segments of kSegIds ids:
//   1. k_bt_exits (one workgroup per segment, all in parallel): for EVERY id k of the segment, where the chain through k
//      leaves the segment (exit = its first ancestor below the segment) and how many of its nodes lie inside it --
//      pointer jumping in LDS on 64-bit (count, next) words, in place: a reader sees the old or the new word of another
//      id, both of which are true statements ("count nodes further on comes `next`"), so no double buffer and no
//      per-round barrier pair are needed, only the convergence vote.
//   2. k_bt_walk (one lane): from the heaviest point, hop from segment to segment over the exits -- n / kSegIds dependent
//      loads instead of one per path point -- recording every visited segment's entry id and output offset.
//   3. k_bt_fill (one lane per segment, all in parallel): the few path points inside the segment, from its entry.
// The path comes out as before: ids in descending order.
constexpr int kSegIds = 16384;        // 128 KiB of LDS: the walk of step 2 is n / kSegIds dependent loads of ~0.3 us
constexpr int kSegThreads = 512;
constexpr uint32_t kSegTerm = 0x80000000u;

__global__ __launch_bounds__(kSegThreads) void k_bt_exits(const int32_t* __restrict__ pred, int64_t n, unsigned long long* __restrict__ ec) {
  extern __shared__ unsigned long long s_w[];                 // [kSegIds] (count << 32) | next: local index, or kSegTerm | (exit id + 1)
  const int tid = threadIdx.x;
  const int64_t lo = (int64_t)blockIdx.x * kSegIds;
  const int m = (int)((n - lo) < kSegIds ? (n - lo) : kSegIds);
  for (int k = tid; k < m; k += kSegThreads) {
    int32_t p = pred[lo + k];
    if ((int64_t)p >= lo + k) p = -1;                         // predecessors have smaller ids: anything else ends the chain
    const uint32_t nxt = (int64_t)p >= lo ? (uint32_t)(p - lo) : (kSegTerm | (uint32_t)(p + 1));
    s_w[k] = (1ull << 32) | nxt;
  }
  __syncthreads();
  while (true) {
    int changed = 0;
    for (int k = tid; k < m; k += kSegThreads) {
      const unsigned long long w = s_w[k];
      const uint32_t nxt = (uint32_t)w;
      if (!(nxt & kSegTerm)) {
        const unsigned long long wj = s_w[nxt];               // nxt < k: another thread's word, read whole (8-byte LDS access)
        s_w[k] = (((w >> 32) + (wj >> 32)) << 32) | (uint32_t)wj;
        changed = 1;
      }
    }
    if (!__syncthreads_or(changed)) break;
  }
  for (int k = tid; k < m; k += kSegThreads) {
    const unsigned long long w = s_w[k];
    ec[lo + k] = (w & 0xFFFFFFFF00000000ull) | (uint32_t)(((uint32_t)w & ~kSegTerm) - 1u);    // exit id (0xFFFFFFFF: none)
  }
}

__global__ __launch_bounds__(64) void k_bt_walk(const unsigned long long* __restrict__ ec, int64_t n, int32_t* __restrict__ seg_entry,
                                                int32_t* __restrict__ seg_base, int64_t* __restrict__ meta, const uint32_t* __restrict__ ctl) {
  if (threadIdx.x != 0) return;
  int32_t cur = (int32_t)meta[0];
  // a column pipeline that gave up (ctl[1], the 20 s neighbour time-out) has left pred[] / meta[0] unwritten: no path
  if ((ctl && ctl[1] != 0u) || (int64_t)cur >= n) cur = -1;
  int64_t total = 0;
  while (cur >= 0) {
    const unsigned long long w = ec[cur];
    const int32_t s = cur / kSegIds;
    seg_entry[s] = cur; seg_base[s] = (int32_t)total;
    total += (int64_t)(w >> 32);
    const int32_t nxt = (int32_t)(uint32_t)w;
    cur = nxt < s * kSegIds ? nxt : -1;                       // an exit lies below its segment: never a cycle, never more than n ids
  }
  meta[1] = total;
}

__global__ __launch_bounds__(64) void k_bt_fill(const int32_t* __restrict__ pred, const int32_t* __restrict__ seg_entry,
                                                const int32_t* __restrict__ seg_base, int64_t n_seg, int32_t* __restrict__ path_ids) {
  const int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= n_seg) return;
  int32_t cur = seg_entry[s];
  if (cur < 0) return;
  const int32_t lo = (int32_t)(s * kSegIds);
  int32_t pos = seg_base[s];
  while (cur >= lo) {
    path_ids[pos++] = cur;
    const int32_t p = pred[cur];
    cur = p < cur ? p : -1;
  }
}

int64_t chain_backtrack_segments(int64_t n) { return (n + kSegIds - 1) / kSegIds; }

// ec: [n] 64-bit words (the column-major quality array is free by now), seg: [2 * segments] int32
static int launch_backtrack(const int32_t* pred, int64_t n, int32_t* path_ids, int64_t* meta, const uint32_t* ctl,
                            unsigned long long* ec, int32_t* seg, hipStream_t s) {
  const int64_t n_seg = chain_backtrack_segments(n);
  if (hipMemsetAsync(seg, 0xFF, sizeof(int32_t) * (size_t)n_seg, s) != hipSuccess) return -1;
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_bt_exits), hipFuncAttributeMaxDynamicSharedMemorySize, kSegIds * 8);
  hipLaunchKernelGGL(k_bt_exits, dim3((unsigned)n_seg), dim3(kSegThreads), kSegIds * 8, s, pred, n, ec);
  hipLaunchKernelGGL(k_bt_walk, dim3(1), dim3(64), 0, s, (const unsigned long long*)ec, n, seg, seg + n_seg, meta, ctl);
  hipLaunchKernelGGL(k_bt_fill, dim3((unsigned)((n_seg + 63) / 64)), dim3(64), 0, s, pred, (const int32_t*)seg, (const int32_t*)(seg + n_seg), n_seg, path_ids);
  return 0;
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

// =====================================================================================================
// Column-pipelined DP (the default).  For a match k of rank column C in audio row i
//     {k' < k, v' <= v} = {matches of columns < C in rows <= i}  u  {matches k' < k of column C with rank' <= rank}
// so a column needs from the columns to its left ONE (sum, id) record per audio row -- B_i(C), the
// lexicographic maximum over their matches in rows <= i -- and hands B_i(C + 1) = max(B_i(C), its own matches
// in rows <= i) to the right.  One single-wavefront workgroup per column; its Fenwick tree covers only the
// column's ranks and lives entirely in LDS; the columns form a pipeline over the rows, 256 rows ("batch")
// at a time.  The columns are contiguous rank ranges of about equal WEIGHT (k_rank_cols below), not equal
// width: the pipeline ends when its slowest column does.  Inside a column the matches are taken 64 at a
// time, one per lane ("window"):
//   * prefix maxima from the tree as it stood before the window (all lanes at once, LDS only);
//   * dominance among the window's own matches by a sequential sweep -- match j's final sum is broadcast
//     (v_readlane) and taken by the later lanes with rank >= rank_j whose best is not larger; later ids win
//     ties, which is what the reference's frontier does (:679-680);
//   * the window's matches enter the tree in two LDS-atomic phases: max on the sums (ds_max_u64: the sums
//     are non-negative doubles, ordered like their bit patterns), then max on the ids where the sum is the
//     lane's own (newer matches have larger ids, so a stale id of a smaller sum always loses).
// Every sum is still "predecessor's sum + q", one IEEE addition, so sums, comparisons and ties are those
// of the reference whatever the column width.
// Hand-over (MI355X: per-XCD L2s are not coherent, a CU's L1 is never refreshed by other CUs' stores): a
// row's record travels as three 8-byte granules {tag, value} -- sum low word, sum high word, id -- each
// written by ONE write-through (sc1) store and read by L1-bypassing (sc1) loads, as relaxed agent-scope
// atomics; the tag (launch salt | batch + 1) IS the ready flag, so there is no counter to poll, no release
// fence and no store drain: the right neighbour re-reads a batch until every tag matches.  A batch's 768
// granules move as twelve coalesced 512-byte accesses and change hands with the lanes that own the rows in
// LDS.  The next batch's granules are requested while the current batch is processed.  Column numbers are taken from a
// ticket counter, so a workgroup only ever waits for workgroups that are already running, and every column
// writes to a buffer of its own (no back-pressure): the pipeline cannot deadlock however many columns are
// resident.  tests/chain_col_model.cpp is the CPU model of exactly this decomposition.

constexpr int kRowsPerLane = 4;                      // a batch = 64 lanes x 4 rows
constexpr int kBatchRows = 64 * kRowsPerLane;
constexpr int kGranules = 3 * kRowsPerLane;          // per lane and batch

struct ColArgs {
  const uint32_t* c_row; const uint16_t* c_lr; const double* c_q; const uint32_t* c_gid;
  const int32_t* col_start; const int32_t* d_nrows;   // *d_nrows = number of audio rows (device)
  const int32_t* col_rank0;                            // [n_cols + 1] first rank of every column
  int n_cols, width;                                   // width: the widest column the launch is sized for
  unsigned long long* msg; int64_t msg_stride;        // [n_cols][msg_stride rows][3] granules
  uint32_t* ctl;
  int32_t* pred; int64_t* meta;
  unsigned long long spin_limit;           // wall-clock ticks (100 MHz) a column may wait for its neighbour
  uint32_t salt;                           // tag = salt << 20 | batch + 1: never equal to what an earlier launch left in the buffer
  unsigned long long* stamps;              // diagnostic build (-DDA_CHAIN_STAMPS): 8 tick counters per column
};
#ifdef DA_CHAIN_STAMPS
#define DA_CSTAMP(k) { const unsigned long long t_ = wall_clock64(); cst[k] += t_ - cst_t; cst_t = t_; }
#else
#define DA_CSTAMP(k)
#endif

namespace {

template <int CTRL, int ROW_MASK>
__device__ __forceinline__ void dpp_fetch(double f, uint32_t id, double& of, uint32_t& oi) {   // lanes outside the pattern read (0, 0)
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(f), CTRL, ROW_MASK, 0xf, true);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(f), CTRL, ROW_MASK, 0xf, true);
  oi = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)id, CTRL, ROW_MASK, 0xf, true);
  of = __hiloint2double(hi, lo);
}
// inclusive prefix maximum over the 64 lanes, lexicographic on (sum, id); (0, 0) is the identity
__device__ __forceinline__ void scan_lexmax(double& f, uint32_t& id) {
  double of; uint32_t oi;
#define DA_SCAN_STEP(CTRL, MASK) dpp_fetch<CTRL, MASK>(f, id, of, oi); if (beats(of, oi, f, id)) { f = of; id = oi; }
  DA_SCAN_STEP(0x111, 0xf)      // row_shr:1
  DA_SCAN_STEP(0x112, 0xf)      // row_shr:2
  DA_SCAN_STEP(0x114, 0xf)      // row_shr:4
  DA_SCAN_STEP(0x118, 0xf)      // row_shr:8   -> prefix within each row of 16 lanes
  DA_SCAN_STEP(0x142, 0xa)      // row_bcast:15 into rows 1 and 3
  DA_SCAN_STEP(0x143, 0xc)      // row_bcast:31 into rows 2 and 3
#undef DA_SCAN_STEP
}
__device__ __forceinline__ void lexmax_into(double& f, uint32_t& id, double of, uint32_t oi) {
  if (beats(of, oi, f, id)) { f = of; id = oi; }
}

}  // namespace

template <int LV>      // LV = longest Fenwick path: width < 2^LV
__global__ __launch_bounds__(64) void k_chain_columns(ColArgs a) {
  extern __shared__ uint4 s_tree[];                  // [0] empty record, [1 .. width] nodes, [width + 1] overflow dummy; then the batch's incoming records and its per-row maxima
  const int lane = threadIdx.x;
  uint4* s_in = s_tree + (a.width + 2);              // [256] records from the left, by row of the batch
  uint4* s_rowmax = s_in + kBatchRows;               // [256] the column's running maximum after each row
  for (int h = lane; h < a.width + 2 + 2 * kBatchRows; h += 64) s_tree[h] = uint4{0u, 0u, 0u, 0u};
  uint32_t col = 0;
  if (lane == 0) col = __hip_atomic_fetch_add(a.ctl, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const int C = (int)__builtin_amdgcn_readfirstlane(col);
  __syncthreads();
  if (C >= a.n_cols) return;
  const int w = a.col_rank0[C + 1] - a.col_rank0[C];     // this column's ranks (<= a.width, the bound LDS is sized for)
  if (w > a.width) {                                     // cannot happen (k_rank_cols); never index LDS past its end
    if (lane == 0) __hip_atomic_store(a.ctl + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return;
  }
  const int32_t n_rows = *a.d_nrows;
  const int32_t n_batches = (n_rows + kBatchRows - 1) / kBatchRows;
  int64_t cursor = a.col_start[C];
  const int64_t end = a.col_start[C + 1];
  const unsigned long long* in8 = C > 0 ? a.msg + (int64_t)(C - 1) * a.msg_stride * 3 : nullptr;
  unsigned long long* out8 = a.msg + (int64_t)C * a.msg_stride * 3;
  const bool has_right = C + 1 < a.n_cols;
  double Mf = 0.0; uint32_t Mid = 0u;               // running maximum over this column's matches (uniform)
  uint32_t last_id = 0u;                            // id of the record of the batch's last row (lane 63)
#ifdef DA_CHAIN_STAMPS
  unsigned long long cst[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long cst_t = wall_clock64();
  // timeline: the 100 MHz wall clock at the column's start ([8 + 31]) and at the end of every `tl_step`-th batch ([8 + k])
  const int32_t tl_step = n_batches > 31 ? (n_batches + 30) / 31 : 1;
  if (lane == 0) a.stamps[(size_t)C * kChainStampWords + 8 + 31] = cst_t;
#endif

  // the window at `cursor`, prefetched: one match per lane
  uint32_t n_row = 0xFFFFFFFFu, n_gid = 0u; uint32_t n_lr = 0u; double n_q = 0.0;
  auto fetch = [&](int64_t at) {
    const int64_t k = at + lane;
    n_row = 0xFFFFFFFFu; n_lr = 0u; n_q = 0.0; n_gid = 0u;
    if (k < end) { n_row = a.c_row[k]; n_lr = a.c_lr[k]; n_q = a.c_q[k]; n_gid = a.c_gid[k]; }
  };
  fetch(cursor);

  // A batch's 768 granules travel coalesced: lane l moves granules 64 g + l (twelve 512-byte accesses per batch; a lane moving
  // its own four rows' twelve granules touched 48 cache lines per instruction and the hand-over took 2.5-4 us of every batch).
  // Granule G is word G % 3 of row G / 3's 16-byte record in LDS, where the lane that owns the row picks it up / puts it down.
  unsigned long long gr[kGranules];
  uint32_t goff[kGranules];
#pragma unroll
  for (int g = 0; g < kGranules; ++g) { const uint32_t G = 64u * (uint32_t)g + (uint32_t)lane; goff[g] = (G / 3u) * 4u + G % 3u; }
  auto request = [&](int32_t b) {
    const unsigned long long* p = in8 + (int64_t)kBatchRows * b * 3 + lane;
#pragma unroll
    for (int g = 0; g < kGranules; ++g) gr[g] = __hip_atomic_load(p + 64 * g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  };
  auto complete = [&](uint32_t tag) -> bool {
    bool ok = true;
#pragma unroll
    for (int g = 0; g < kGranules; ++g) ok &= (uint32_t)(gr[g] >> 32) == tag;
    return __all(ok);
  };
  if (C > 0 && n_batches > 0) request(0);

  for (int32_t b = 0; b < n_batches; ++b) {
    DA_CSTAMP(6)
    const uint32_t tag = (a.salt << 20) | (uint32_t)(b + 1);
    // ---- the records of this batch's rows from the left neighbour: wait until every granule carries the tag
    if (C > 0) {
      unsigned spins = 0;
      const unsigned long long t_wait = wall_clock64();
      while (!complete(tag)) {
        __builtin_amdgcn_s_sleep(1);
        if ((++spins & 255u) == 0u) {
          const uint32_t ab = __builtin_amdgcn_readfirstlane(__hip_atomic_load(a.ctl + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
          if (ab != 0u) return;
          if (wall_clock64() - t_wait > a.spin_limit) {
            if (lane == 0) __hip_atomic_store(a.ctl + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return;
          }
        }
        request(b);
      }
    }
    if (C > 0) {
      uint32_t* w_in = reinterpret_cast<uint32_t*>(s_in);      // the records' fourth words stay zero
#pragma unroll
      for (int g = 0; g < kGranules; ++g) w_in[goff[g]] = (uint32_t)gr[g];
      __syncthreads();
    }
    double Bf[kRowsPerLane]; uint32_t Bid[kRowsPerLane];
#pragma unroll
    for (int r = 0; r < kRowsPerLane; ++r) {
      const uint4 rec = s_in[kRowsPerLane * lane + r];          // column 0: zeros, the empty record
      Bf[r] = node_cum(rec); Bid[r] = rec.z;
      s_rowmax[kRowsPerLane * lane + r] = uint4{0u, 0u, 0u, 0u};
    }
    if (C > 0 && b + 1 < n_batches) request(b + 1);       // the next batch, behind this one's work
    DA_CSTAMP(1)
    const double Mstart_f = Mf; const uint32_t Mstart_id = Mid;
    const uint32_t row_base = (uint32_t)(kBatchRows * b);
    const uint32_t row_end = row_base + (uint32_t)kBatchRows;

    while (true) {
      // ---- this window: the matches at `cursor` that belong to the batch (a prefix of the lanes)
      const uint32_t row = n_row; const uint32_t gid = n_gid; const double qv0 = n_q; const uint32_t lr0 = n_lr;
      const bool inb = row < row_end;                 // lanes past the column's end carry row = 0xFFFFFFFF
      const int cnt = __popcll(__ballot(inb));
      if (cnt == 0) break;
      cursor += cnt;
      fetch(cursor);                                  // the next window, behind this one's work
      const uint32_t lr = inb ? lr0 : 0u;             // rank 0: the empty record, on every path
      const double qv = inb ? qv0 : 0.0;
      const uint32_t id1 = inb ? gid + 1u : 0u;
      const int rl = inb ? (int)(row - row_base) : 0;
      const uint4 bp = s_in[rl];
      const double Bpf = node_cum(bp);
      const uint32_t Bpid = bp.z;

      // ---- prefix maximum over the column's earlier windows: Fenwick query, all lanes at once
      double tf = 0.0; uint32_t tid = 0u;
      {
        uint4 nd[LV];
        uint32_t x = lr;
#pragma unroll
        for (int l = 0; l < LV; ++l) { nd[l] = s_tree[x]; x &= x - 1u; }
#pragma unroll
        for (int l = 0; l < LV; ++l) tf = max_f64(tf, node_cum(nd[l]));
#pragma unroll
        for (int l = 0; l < LV; ++l) tid = (node_cum(nd[l]) == tf && nd[l].z > tid) ? nd[l].z : tid;
      }
#ifdef DA_CHAIN_STAMPS
      if (tf < 0.0) return;
#endif
      DA_CSTAMP(2)
      // ---- dominance inside the window: match j's final sum goes to the later lanes it precedes
      double gcol = tf; int winj = -1;
      // (A hand-scheduled version of this loop -- exec-masked v_max_f64, ten vector instructions per match --
      // measured 115 cycles per match against 103 for the compiler's: the step is bound by the dependent
      // chain add -> v_readlane -> SGPR -> max, 83 cycles by itself, not by instruction count.)
      for (int j = 0; j + 1 < cnt; ++j) {
        const double fcur = qv + max_f64(gcol, Bpf);
        const double fj = read_lane(fcur, j);
        const uint32_t lrj = (uint32_t)__builtin_amdgcn_readlane((int)lr, j);
        const bool take = lane > j && lr >= lrj && fj >= gcol;
        gcol = take ? fj : gcol; winj = take ? j : winj;
      }
      const double F = inb ? qv + max_f64(gcol, Bpf) : 0.0;      // idle lanes: the identity record, on the empty path
      DA_CSTAMP(3)
      {
        const uint32_t wid = (uint32_t)__shfl((int)id1, winj < 0 ? 0 : winj);
        const uint32_t cid = winj >= 0 ? wid : tid;   // the column's own candidate (sum gcol)
        const uint32_t pid = beats(Bpf, Bpid, gcol, cid) ? Bpid : cid;
        if (inb) a.pred[gid] = (int32_t)pid - 1;
      }
      // ---- the window's matches enter the tree: sums first, then ids where the sum is this lane's
      {
        const unsigned long long Fb = (unsigned long long)__double_as_longlong(F);
        uint32_t x = lr;
        uint32_t path[LV];
        // lanes whose path has left the column (and idle lanes: rank 0) sit the level out -- sent to a dummy node instead they
        // all hit ONE LDS address and the atomic unit takes them one after the other
        if (!inb) x = (uint32_t)w + 1u;
#pragma unroll
        for (int l = 0; l < LV; ++l) {
          path[l] = x <= (uint32_t)w ? x : 0u;
          if (path[l] != 0u) atomicMax(reinterpret_cast<unsigned long long*>(&s_tree[path[l]]), Fb);
          x += x & (0u - x);
        }
#pragma unroll
        for (int l = 0; l < LV; ++l) {
          if (path[l] != 0u) {
            const unsigned long long cur = __hip_atomic_load(reinterpret_cast<unsigned long long*>(&s_tree[path[l]]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (cur == Fb) atomicMax(reinterpret_cast<unsigned int*>(&s_tree[path[l]]) + 2, id1);
          }
        }
      }
      DA_CSTAMP(4)
      // ---- running maximum of the column after every row of the window
      double rf = F; uint32_t rid = id1;
      scan_lexmax(rf, rid);
      lexmax_into(rf, rid, Mf, Mid);
      const uint32_t nrow = (uint32_t)__shfl_down((int)row, 1);
      if (inb && (lane == cnt - 1 || nrow != row)) s_rowmax[rl] = make_node(rf, rid);
      Mf = read_lane(rf, cnt - 1); Mid = (uint32_t)__builtin_amdgcn_readlane((int)rid, cnt - 1);
      DA_CSTAMP(5)
#ifdef DA_CHAIN_STAMPS
      cst[7] += 1;
#endif
      if (cnt < 64) break;
    }
    // ---- records for the right neighbour: max(incoming, the column's maximum over rows <= this one).
    // Rows without a match of this column inherit the previous row's maximum: a prefix maximum over the
    // rows (serial over the lane's four, then across the lanes).
    {
      double rf[kRowsPerLane]; uint32_t ri[kRowsPerLane];
#pragma unroll
      for (int r = 0; r < kRowsPerLane; ++r) {
        const uint4 rm = s_rowmax[kRowsPerLane * lane + r];
        rf[r] = node_cum(rm); ri[r] = rm.z;
        if (r > 0) lexmax_into(rf[r], ri[r], rf[r - 1], ri[r - 1]);
      }
      double tf = rf[kRowsPerLane - 1]; uint32_t ti = ri[kRowsPerLane - 1];
      scan_lexmax(tf, ti);                            // inclusive over the lanes' totals
      double ef; uint32_t ei;
      dpp_fetch<0x138, 0xf>(tf, ti, ef, ei);          // wave_shr:1 -> exclusive (lane 0 reads the identity)
      lexmax_into(ef, ei, Mstart_f, Mstart_id);
#pragma unroll
      for (int r = 0; r < kRowsPerLane; ++r) {
        lexmax_into(rf[r], ri[r], ef, ei);
        lexmax_into(rf[r], ri[r], Bf[r], Bid[r]);
        if (has_right) s_rowmax[kRowsPerLane * lane + r] = make_node(rf[r], ri[r]);    // staged for the coalesced stores
      }
      if (has_right) {
        __syncthreads();
        const uint32_t* w_out = reinterpret_cast<const uint32_t*>(s_rowmax);
        const unsigned long long hi = (unsigned long long)tag << 32;
        unsigned long long* p = out8 + (int64_t)kBatchRows * b * 3 + lane;
#pragma unroll
        for (int g = 0; g < kGranules; ++g) __hip_atomic_store(p + 64 * g, hi | w_out[goff[g]], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      last_id = ri[kRowsPerLane - 1];
    }
#ifdef DA_CHAIN_STAMPS
    if (lane == 0 && b % tl_step == 0 && b / tl_step < 31) a.stamps[(size_t)C * kChainStampWords + 8 + b / tl_step] = wall_clock64();
#endif
  }
#ifdef DA_CHAIN_STAMPS
  DA_CSTAMP(6)
  if (lane == 0) for (int k = 0; k < 8; ++k) a.stamps[(size_t)C * kChainStampWords + k] = cst[k];
#endif
  // the last column's record of the last row is the heaviest match overall (rows past the end repeat it)
  if (C + 1 == a.n_cols && lane == 63) { a.meta[0] = (int64_t)last_id - 1; a.meta[1] = 0; }
}

// ---- preparation of the column-major arrays
struct U8ToI32 { __device__ int32_t operator()(const uint8_t& x) const { return (int32_t)x; } };

// ---- columns of equal WEIGHT instead of equal width.  The pipeline finishes when its slowest column does, and a column's
// time goes with its matches: measured on the 2 h pair with equal widths the columns' busy time ran from 10 to 33 ms around a
// mean of 21.5, and the pipeline (50 ms) was the heaviest early column plus the drain behind it.  A rank weighs its matches
// plus the average number of matches per rank, so a column holds at most twice the average number of ranks (the LDS tree is
// sized for that) and about 1 / n_cols of the weight.  hist -> exclusive sums -> column of every rank -> first rank of
// every column; all on the device, nothing comes back to the host.
// (every `stride`-th match only: the weights steer a partition, they are not part of the result, and 7.2e7 device-scope
// atomics took 3.5 ms)
__global__ __launch_bounds__(256) void k_rank_hist(const int32_t* __restrict__ rank, int64_t n, int64_t stride, int32_t* __restrict__ hist) {
  const int64_t k = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * stride;
  if (k < n) atomicAdd(&hist[rank[k]], 1);            // ranks are 1 .. n_ranks (k_chain_prep clamps what the rank map does not know)
}

// excl[r] = matches with rank < r  ->  col[r] in place; col[0] = -1, col[n_ranks + 1] = n_cols
__global__ __launch_bounds__(256) void k_rank_cols(int32_t* __restrict__ excl, int64_t n_ranks, int64_t n, int n_cols) {
  const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r > n_ranks + 1) return;
  int32_t c;
  if (r == 0) c = -1;
  else if (r == n_ranks + 1) c = n_cols;
  else {
    // weight in front of rank r, in units of 1 / n_ranks matches: its predecessors' matches + (r - 1) average shares
    const double wfront = (double)excl[r] * (double)n_ranks + (double)(r - 1) * (double)n;
    const double wtotal = 2.0 * (double)n * (double)n_ranks;
    c = (int32_t)(wfront / wtotal * (double)n_cols);  // monotone in r: the columns are contiguous rank ranges
    c = c < 0 ? 0 : (c >= n_cols ? n_cols - 1 : c);
  }
  excl[r] = c;
}

__global__ __launch_bounds__(256) void k_col_bounds(const int32_t* __restrict__ rcol, int64_t n_ranks, int32_t* __restrict__ col_rank0) {
  const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x + 1;
  if (r > n_ranks + 1) return;
  for (int c = rcol[r - 1] + 1; c <= rcol[r]; ++c) col_rank0[c] = (int32_t)r;   // first rank of every column (empty ones: width 0)
}

__global__ __launch_bounds__(256) void k_col_keys(const int32_t* __restrict__ rank, int64_t n, const int32_t* __restrict__ rcol, int n_cols,
                                                  uint16_t* __restrict__ key, uint32_t* __restrict__ val) {
  const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n) return;
  uint32_t col = (uint32_t)rcol[rank[k]];
  if (col >= (uint32_t)n_cols) col = (uint32_t)n_cols - 1u;          // belt and braces: k_col_gather indexes col_start[] by this
  key[k] = (uint16_t)col;
  val[k] = (uint32_t)k;
}

__global__ __launch_bounds__(256) void k_col_gather(const uint16_t* __restrict__ key, const uint32_t* __restrict__ val, int64_t n,
                                                    const int32_t* __restrict__ col_rank0, int n_cols, const int32_t* __restrict__ rank,
                                                    const double* __restrict__ q, const int32_t* __restrict__ rowid1, uint32_t* __restrict__ c_row,
                                                    uint16_t* __restrict__ c_lr, double* __restrict__ c_q, uint32_t* __restrict__ c_gid,
                                                    int32_t* __restrict__ col_start) {
  const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j > n) return;
  const int prev = j > 0 ? (int)key[j - 1] : -1;
  const int cur = j < n ? ((int)key[j] < n_cols ? (int)key[j] : n_cols - 1) : n_cols;
  for (int c = prev + 1; c <= cur; ++c) col_start[c] = (int32_t)j;      // first match of every column (empty ones included)
  if (j == n) return;
  const uint32_t g = val[j];
  c_row[j] = (uint32_t)(rowid1[g] - 1);
  c_lr[j] = (uint16_t)(rank[g] - col_rank0[cur] + 1);
  c_q[j] = q[g];
  c_gid[j] = g;
}

namespace {
int bit_length(int64_t x) { int b = 0; while (((int64_t)1 << b) <= x) ++b; return b; }
int col_bits(int n_cols) { return std::max(1, bit_length((int64_t)n_cols - 1)); }
constexpr int kColMaxWidth = 8191;      // LV 13: 128 KiB of tree
constexpr int kColMaxCols = 4096;
}  // namespace

ChainColumnPlan chain_columns_plan(int64_t n, int64_t n_ranks, int64_t max_cols) {
  // More columns = more wavefronts working and a shorter serial sweep per column, but every column walks all the batches
  // and the pipeline takes n_cols x (batch time + hand-over latency) to fill and to drain (a cell waits for its left
  // neighbour and its predecessor in the column: the DP's time is the longest dependent path through the (column, batch)
  // grid, profiles/tools/chain_cells.py).  Measured (MI355X, round 4 kernel): 2 h pair, 7.2e7 matches: 448 columns 42.4 ms,
  // 512: 41.3, 640: 39.8, 768: 38.4, 1024: 37.7, 1280: 37.8, 1536: 38.9; 22 min pair, 3.1e6 matches: 96: 7.4, 127: 6.8,
  // 192: 6.3, 256: 6.2, 320: 6.5.  About 12 k matches per column, at most 1 024.  `width` is the bound the launch is sized
  // for: a column holds at most twice the average number of ranks (see k_rank_cols) and must fit LDS.
  if (n_ranks < 1) n_ranks = 1;
  int64_t nc = n / 12288;
  if (const char* e = std::getenv("DALIGN_CHAIN_COLS")) nc = std::atoll(e);
  else nc = std::min<int64_t>(nc, n >= 300000000LL ? 2048 : 1024);   // 8 h pair, 1.12e9 matches, 2.5e6 ranks: 1 024 columns 907 ms (86 KB of LDS each: one per CU, four rounds), 2 048: 363 ms
  if (max_cols > 0) nc = std::min<int64_t>(nc, max_cols);             // the caller's memory budget for the hand-over records
  nc = std::min<int64_t>(nc, (n_ranks + 63) / 64);
  {
    const int64_t avg_max = (kColMaxWidth - 2) / 2;                   // widest column = 2 x average + 2 must fit the LDS tree
    nc = std::max<int64_t>(nc, (n_ranks + avg_max - 1) / avg_max);
  }
  nc = std::max<int64_t>(1, std::min<int64_t>(nc, kColMaxCols));
  const int64_t avg = (n_ranks + nc - 1) / nc;
  const int64_t width = nc == 1 ? n_ranks : std::min<int64_t>(n_ranks, 2 * avg + 2);   // + 1 for the rounding of k_rank_cols' quotient
  return ChainColumnPlan{(int)nc, (int)width};
}

size_t chain_columns_lds_bytes(int width) { return (size_t)(width + 2 + 2 * kBatchRows) * 16; }
int chain_columns_batch_rows() { return kBatchRows; }

size_t chain_columns_temp_bytes(int64_t n, int64_t n_ranks) {
  size_t b1 = 0, b2 = 0;
  const int m = (int)std::max<int64_t>(std::max<int64_t>(1, n), n_ranks + 2);
  (void)hipcub::DeviceRadixSort::SortPairs(nullptr, b1, (const uint16_t*)nullptr, (uint16_t*)nullptr, (const uint32_t*)nullptr, (uint32_t*)nullptr, m, 0, 16);
  hipcub::TransformInputIterator<int32_t, U8ToI32, const uint8_t*> it((const uint8_t*)nullptr, U8ToI32{});
  (void)hipcub::DeviceScan::InclusiveSum(nullptr, b2, it, (int32_t*)nullptr, m);
  return std::max(b1, b2);
}

int launch_chain_columns(const ChainLaunch& c, const ChainColumns& cc, hipStream_t s) {
  if (c.n <= 0) return 0;
  if (c.n > 0x7fffffffLL || cc.n_cols < 1 || cc.n_cols > kColMaxCols || cc.width < 1 || cc.width > kColMaxWidth ||
      (int64_t)cc.n_cols * cc.width < c.n_ranks || c.n_ranks + 2 > 0x7fffffffLL) return -2;
  const int n = (int)c.n;
  const unsigned blocks = (unsigned)((c.n + 256) / 256);               // covers j = n as well
  // dense row ordinals: inclusive scan of the row-head flags
  {
    size_t bytes = cc.temp_bytes;
    hipcub::TransformInputIterator<int32_t, U8ToI32, const uint8_t*> it(c.flags, U8ToI32{});
    if (hipcub::DeviceScan::InclusiveSum(cc.temp, bytes, it, cc.rowid1, n, s) != hipSuccess) return -3;
  }
  // the columns' rank ranges: equal weight (matches + an average share per rank)
  {
    const int64_t m = c.n_ranks + 2;
    int32_t* hist = cc.rank_cum; int32_t* rcol = cc.rank_cum + m;
    if (hipMemsetAsync(hist, 0, sizeof(int32_t) * (size_t)m, s) != hipSuccess) return -4;
    const int64_t stride = std::max<int64_t>(1, c.n >> 22);          // ~4e6 samples
    const int64_t n_samples = (c.n + stride - 1) / stride;
    hipLaunchKernelGGL(k_rank_hist, dim3((unsigned)((n_samples + 255) / 256)), dim3(256), 0, s, c.rank, c.n, stride, hist);
    size_t bytes = cc.temp_bytes;
    if (hipcub::DeviceScan::ExclusiveSum(cc.temp, bytes, hist, rcol, (int)m, s) != hipSuccess) return -5;
    const unsigned rb = (unsigned)((m + 255) / 256);
    hipLaunchKernelGGL(k_rank_cols, dim3(rb), dim3(256), 0, s, rcol, c.n_ranks, n_samples, cc.n_cols);
    hipLaunchKernelGGL(k_col_bounds, dim3(rb), dim3(256), 0, s, rcol, c.n_ranks, cc.col_rank0);
  }
  const int32_t* rcol = cc.rank_cum + (c.n_ranks + 2);
  // stable partition of the match ids by column
  hipLaunchKernelGGL(k_col_keys, dim3(blocks), dim3(256), 0, s, c.rank, c.n, rcol, cc.n_cols, cc.key_in, cc.val_in);
  {
    size_t bytes = cc.temp_bytes;
    if (hipcub::DeviceRadixSort::SortPairs(cc.temp, bytes, (const uint16_t*)cc.key_in, cc.key_out, (const uint32_t*)cc.val_in, cc.val_out, n, 0,
                                           col_bits(cc.n_cols), s) != hipSuccess) return -6;
  }
  hipLaunchKernelGGL(k_col_gather, dim3(blocks), dim3(256), 0, s, cc.key_out, cc.val_out, c.n, cc.col_rank0, cc.n_cols, c.rank, c.q, cc.rowid1,
                     cc.c_row, cc.c_lr, cc.c_q, cc.c_gid, cc.col_start);
  if (hipMemsetAsync(cc.ctl, 0, sizeof(uint32_t) * (size_t)(kChainCtlHead + cc.n_cols), s) != hipSuccess) return -7;
  ColArgs a{};
  a.c_row = cc.c_row; a.c_lr = cc.c_lr; a.c_q = cc.c_q; a.c_gid = cc.c_gid; a.col_start = cc.col_start;
  a.d_nrows = cc.rowid1 + (c.n - 1);
  a.col_rank0 = cc.col_rank0;
  a.n_cols = cc.n_cols; a.width = cc.width; a.msg = cc.msg; a.msg_stride = cc.msg_stride; a.ctl = cc.ctl; a.salt = cc.salt;
  a.pred = c.pred; a.meta = c.meta;
  {                                                                    // 20 s of the 100 MHz wall clock (DALIGN_CHAIN_SPIN_SECONDS: stress runs with thousands of forced columns beside a GEMM that never lets go of the CUs)
    unsigned long long secs = 20ull;
    if (const char* e = std::getenv("DALIGN_CHAIN_SPIN_SECONDS")) secs = (unsigned long long)std::max(1ll, std::atoll(e));
    a.spin_limit = 100000000ull * secs;
  }
  a.stamps = reinterpret_cast<unsigned long long*>(cc.ctl + kChainCtlHead + ((cc.n_cols + 1) & ~1));
  const size_t lds = chain_columns_lds_bytes(cc.width);
  const int lv = bit_length(cc.width);
  auto go = [&](auto kernel) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(kernel, dim3((unsigned)cc.n_cols), dim3(64), lds, s, a);
  };
  if (lv <= 7) go(k_chain_columns<7>);
  else if (lv <= 9) go(k_chain_columns<9>);
  else if (lv <= 10) go(k_chain_columns<10>);
  else if (lv <= 11) go(k_chain_columns<11>);
  else if (lv <= 12) go(k_chain_columns<12>);
  else go(k_chain_columns<13>);
  if (launch_backtrack(c.pred, c.n, c.path_ids, c.meta, (const uint32_t*)cc.ctl, c.bt_ec, c.bt_seg, s) != 0) return -8;
  hipLaunchKernelGGL(k_chain_gather, dim3(256), dim3(256), 0, s, c.keys, c.path_ids, c.meta, c.out_i, c.out_v);
  return 0;
}

int chain_tree_shift(int64_t n_ranks) {
  // LDS holds the levels with span >= 2^S: (n_ranks >> S) + 1 nodes of 16 B within 128 KiB
  int S = 6;
  while (S < kLowMax && ((n_ranks >> S) + 2 + kScrap) * 16 > 128 * 1024) ++S;
  return S;
}

size_t chain_rows_temp_bytes(int64_t n) {
  size_t bytes = 0;
  hipcub::CountingInputIterator<int32_t> ids(0);
  (void)hipcub::DeviceSelect::Flagged(nullptr, bytes, ids, (const uint8_t*)nullptr, (int32_t*)nullptr, (int32_t*)nullptr, (int)n);
  return bytes;
}

// per-match ranks, row-head flags, validation (reads the rank map: runs where that is built)
int launch_chain_prep(const ChainLaunch& c, hipStream_t s, bool columns) {
  if (c.n > 0x7fffffffLL || c.n_ranks >= (1LL << 24)) return -1;
  const int S = chain_tree_shift(c.n_ranks);
  if (!columns && ((c.n_ranks >> S) + 2 + kScrap) * 16 > 150 * 1024) return -1;    // the one-workgroup kernels keep the upper tree levels in LDS
  if (c.n <= 0) return 0;
  const unsigned blocks = (unsigned)((c.n + 255) / 256);
  hipLaunchKernelGGL(k_chain_prep, dim3(blocks), dim3(256), 0, s, c.keys, c.q, c.n, c.rankmap, c.rankmap_len, c.dense, c.rank, c.flags, c.err);
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
  a.xcd = c.xcd; a.claim = c.d_nrows + 24;         // two ints of the slot's `small` block, zeroed with it
  a.n_ranks = c.n_ranks; a.S = S; a.tree_lo = reinterpret_cast<uint4*>(c.tree_lo); a.pred = c.pred; a.meta = c.meta;
  const size_t lds = (size_t)((c.n_ranks >> S) + 2 + kScrap) * 16;
  int hbits = 0;
  while (((int64_t)1 << hbits) <= (c.n_ranks >> S)) ++hbits;          // levels held in LDS
  auto go = [&](auto kernel) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(kernel, dim3(c.xcd >= 0 ? 8 : 1), dim3(64), lds, s, a);
  };
  const int hsel = hbits <= 10 ? 10 : (hbits <= 13 ? 13 : 16);
  if (c.wide) {
    auto go4 = [&](auto kernel) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      hipLaunchKernelGGL(kernel, dim3(c.xcd >= 0 ? 8 : 1), dim3(64 * kW4), lds, s, a);
    };
#define DA_CHAIN_CASE4(L, Hh) if (S == L && hsel == Hh) go4(k_chain_forward_w4<L, Hh>);
    DA_CHAIN_CASE4(6, 10) DA_CHAIN_CASE4(6, 13) DA_CHAIN_CASE4(6, 16)
    DA_CHAIN_CASE4(7, 10) DA_CHAIN_CASE4(7, 13) DA_CHAIN_CASE4(7, 16)
    DA_CHAIN_CASE4(8, 10) DA_CHAIN_CASE4(8, 13) DA_CHAIN_CASE4(8, 16)
#undef DA_CHAIN_CASE4
  } else {
#define DA_CHAIN_CASE(L, Hh) if (S == L && hsel == Hh) go(k_chain_forward<L, Hh>);
  DA_CHAIN_CASE(6, 10) DA_CHAIN_CASE(6, 13) DA_CHAIN_CASE(6, 16)
  DA_CHAIN_CASE(7, 10) DA_CHAIN_CASE(7, 13) DA_CHAIN_CASE(7, 16)
  DA_CHAIN_CASE(8, 10) DA_CHAIN_CASE(8, 13) DA_CHAIN_CASE(8, 16)
#undef DA_CHAIN_CASE
  }
  if (launch_backtrack(c.pred, c.n, c.path_ids, c.meta, nullptr, c.bt_ec, c.bt_seg, s) != 0) return -1;
  hipLaunchKernelGGL(k_chain_gather, dim3(256), dim3(256), 0, s, c.keys, c.path_ids, c.meta, c.out_i, c.out_v);
  return 0;
}

// number of distinct audio rows of a sorted key list (i << 32 | v): one count per block, one atomic each
__global__ __launch_bounds__(256) void k_count_rows(const unsigned long long* __restrict__ keys, int64_t n, unsigned long long* __restrict__ count) {
  __shared__ int s_part[4];
  int mine = 0;
  for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += (int64_t)gridDim.x * blockDim.x)
    mine += (k == 0 || (uint32_t)(keys[k - 1] >> 32) != (uint32_t)(keys[k] >> 32)) ? 1 : 0;
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) mine += __shfl_down(mine, d);
  if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = mine;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(count, (unsigned long long)(s_part[0] + s_part[1] + s_part[2] + s_part[3]));
}
void launch_count_rows(const unsigned long long* keys, int64_t n, unsigned long long* d_count, hipStream_t s) {
  if (n <= 0) return;
  const unsigned blocks = (unsigned)std::min<int64_t>((n + 255) / 256, 4096);
  hipLaunchKernelGGL(k_count_rows, dim3(blocks), dim3(256), 0, s, keys, n, d_count);
}

// ---- dense ranks: the video frames that occur in a sorted match list, numbered in order.  The row list of the match
// stage names every non-quiet frame (2 h pair: 3.6e5), matches exist for fewer than half of them (1.6e5): trees, LDS and
// the number of columns an 8 h pair needs (6e6 listed frames do not fit the chip's LDS at once) go with the smaller number.
__global__ __launch_bounds__(256) void k_mark_used(const unsigned long long* __restrict__ keys, int64_t n, int64_t lv, int32_t* __restrict__ used) {
  const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n) return;
  const uint32_t v = (uint32_t)keys[k];
  if ((int64_t)v < lv) used[v] = 1;                       // every writer stores the same value
}
size_t dense_ranks_temp_bytes(int64_t lv) {
  size_t b = 0;
  (void)hipcub::DeviceScan::ExclusiveSum(nullptr, b, (const int32_t*)nullptr, (int32_t*)nullptr, (int)(lv + 1));
  return b;
}
// used / excl: [lv + 1] each; excl[v] = frames < v that have a match, excl[lv] = their number
int launch_dense_ranks(const unsigned long long* keys, int64_t n, int64_t lv, int32_t* used, int32_t* excl, void* temp, size_t temp_bytes, hipStream_t s) {
  if (lv + 1 > 0x7fffffffLL) return -1;
  if (hipMemsetAsync(used, 0, sizeof(int32_t) * (size_t)(lv + 1), s) != hipSuccess) return -1;
  if (n > 0) hipLaunchKernelGGL(k_mark_used, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, keys, n, lv, used);
  size_t bytes = temp_bytes;
  if (hipcub::DeviceScan::ExclusiveSum(temp, bytes, (const int32_t*)used, excl, (int)(lv + 1), s) != hipSuccess) return -1;
  return 0;
}

void launch_rankmap(const int32_t* vlist, int64_t n_v, int32_t* rankmap, hipStream_t s) {
  if (n_v <= 0) return;
  hipLaunchKernelGGL(k_rankmap, dim3((unsigned)((n_v + 255) / 256)), dim3(256), 0, s, vlist, n_v, rankmap);
}

}  // namespace da
