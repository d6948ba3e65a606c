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
int launch_chain_prep(const ChainLaunch& c, hipStream_t s) {
  if (c.n > 0x7fffffffLL || c.n_ranks >= (1LL << 24)) return -1;
  const int S = chain_tree_shift(c.n_ranks);
  if (((c.n_ranks >> S) + 2 + kScrap) * 16 > 150 * 1024) return -1;
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
  hipLaunchKernelGGL(k_chain_backtrack, dim3(1), dim3(256), 0, s, c.pred, c.n, c.path_ids, c.meta);
  hipLaunchKernelGGL(k_chain_gather, dim3(256), dim3(256), 0, s, c.keys, c.path_ids, c.meta, c.out_i, c.out_v);
  return 0;
}

void launch_rankmap(const int32_t* vlist, int64_t n_v, int32_t* rankmap, hipStream_t s) {
  if (n_v <= 0) return;
  hipLaunchKernelGGL(k_rankmap, dim3((unsigned)((n_v + 255) / 256)), dim3(256), 0, s, vlist, n_v, rankmap);
}

}  // namespace da
