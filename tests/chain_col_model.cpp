// CPU model of the column-pipelined chain DP (describealign_amd/csrc/dalign_chain.hip, k_chain_columns).
// Test infrastructure: tests/test_host_cpu.py compiles this file with g++ and checks the model against the
// host utility (da_chain with a NULL context) on random, tie-heavy instances, so the DECOMPOSITION the
// kernel uses -- rank columns, 256-row record batches, 64-point windows, two-phase tree update -- is
// pinned on the CPU; the GPU tests then pin the kernel itself.
//
// Recurrence (describealign.py:654-656, :674-697): over matches sorted by (audio frame i, video frame v)
//     f[k] = q[k] + max{ f[k'] : k' < k, v[k'] <= v[k] },  ties on f resolved to the larger k'.
// Decomposition: video ranks are cut into columns -- `w` > 0: columns of w ranks; `w` < 0: -w columns of about equal
// weight, by the kernel's own formula (k_rank_cols: a rank weighs its points + the average number of points per rank,
// the column of a rank is floor(weight in front of it / total weight * columns)).  For a point k of column C in row i
//     {k' < k, v' <= v} = {points of columns < C in rows <= i}  u  {points k' < k of column C with rank' <= rank}
// so column C needs from the columns to its left ONE (sum, id) record per audio row -- B_i(C), the
// lexicographic maximum over their points in rows <= i -- and hands B_i(C+1) = max(B_i(C), its own points
// in rows <= i) to the right.  Columns therefore form a pipeline over the rows; inside a column, points are
// taken 64 at a time (a "window"): prefix maxima from a Fenwick tree over the column's ranks as it stood
// before the window, dominance among the window's own points by a 64-step sequential sweep, then the
// window's points enter the tree.
#include <algorithm>
#include <cstdint>
#include <cstring>
#include <vector>

namespace {

struct Rec { double f; uint32_t id1; };                      // id1 = id + 1, 0 = none
inline bool beats(const Rec& a, const Rec& b) { return a.f > b.f || (a.f == b.f && a.id1 > b.id1); }
inline Rec lexmax(const Rec& a, const Rec& b) { return beats(b, a) ? b : a; }
constexpr int kBatchRows = 256;                               // rows per hand-over batch (64 lanes x 4 rows in the kernel)

}  // namespace

extern "C" int chain_col_model(const int32_t* pi, const int32_t* pv, const double* pq, int64_t n, int w,
                               int32_t* pred, int64_t* best) {
  *best = -1;
  if (n <= 0) return 0;
  // dense 1-based video ranks, dense row ordinals
  int32_t vmax = 0;
  for (int64_t k = 0; k < n; ++k) vmax = std::max(vmax, pv[k]);
  std::vector<int32_t> rk((size_t)vmax + 2, 0);
  for (int64_t k = 0; k < n; ++k) rk[pv[k]] = 1;
  int32_t n_ranks = 0;
  for (size_t x = 0; x < rk.size(); ++x) if (rk[x]) rk[x] = ++n_ranks;
  std::vector<int32_t> rowid((size_t)n);
  int32_t n_rows = 0;
  for (int64_t k = 0; k < n; ++k) { if (k == 0 || pi[k] != pi[k - 1]) ++n_rows; rowid[k] = n_rows - 1; }
  // column of every rank (1-based ranks) and first rank of every column
  std::vector<int32_t> rcol((size_t)n_ranks + 2, 0);
  int NC;
  if (w > 0) {
    NC = (n_ranks + w - 1) / w;
    for (int32_t r = 1; r <= n_ranks; ++r) rcol[r] = (r - 1) / w;
  } else {
    NC = -w;
    std::vector<int64_t> hist((size_t)n_ranks + 2, 0);
    for (int64_t k = 0; k < n; ++k) ++hist[rk[pv[k]]];
    int64_t front = 0;
    for (int32_t r = 1; r <= n_ranks; ++r) {
      const double wfront = (double)front * (double)n_ranks + (double)(r - 1) * (double)n;
      const double wtotal = 2.0 * (double)n * (double)n_ranks;
      rcol[r] = std::max(0, std::min(NC - 1, (int)(wfront / wtotal * (double)NC)));
      front += hist[r];
    }
  }
  std::vector<int32_t> rank0((size_t)NC + 1, n_ranks + 1);
  for (int32_t r = n_ranks; r >= 1; --r) rank0[rcol[r]] = r;
  for (int c = NC - 1; c >= 0; --c) if (rank0[c] > rank0[c + 1]) rank0[c] = rank0[c + 1];   // empty columns: width 0
  int wmax = 1;
  for (int c = 0; c < NC; ++c) wmax = std::max(wmax, rank0[c + 1] - rank0[c]);
  const int n_batches = (n_rows + kBatchRows - 1) / kBatchRows;
  // stable partition by column
  std::vector<std::vector<int32_t>> cols((size_t)NC);
  for (int64_t k = 0; k < n; ++k) cols[rcol[rk[pv[k]]]].push_back((int32_t)k);
  std::vector<Rec> Bin((size_t)n_batches * kBatchRows, Rec{0.0, 0u}), Bout((size_t)n_batches * kBatchRows);
  int LV = 1; while ((1 << LV) <= wmax) ++LV;                  // longest Fenwick path (the launch is sized for the widest column)
  for (int C = 0; C < NC; ++C) {
    const std::vector<int32_t>& P = cols[C];
    const int w = rank0[C + 1] - rank0[C];                     // this column's ranks (shadows the argument)
    std::vector<Rec> tree((size_t)w + 2, Rec{0.0, 0u});        // [0] empty record
    Rec M{0.0, 0u};
    size_t cursor = 0;
    for (int b = 0; b < n_batches; ++b) {
      Rec rowmax[kBatchRows];
      for (int r = 0; r < kBatchRows; ++r) rowmax[r] = Rec{0.0, 0u};
      const Rec Mstart = M;
      while (true) {
        int cnt = 0;
        while (cnt < 64 && cursor + cnt < P.size() && rowid[P[cursor + cnt]] < kBatchRows * (b + 1)) ++cnt;
        if (cnt == 0) break;
        int32_t row[64], lr[64], gid[64]; double q[64]; Rec B[64];
        for (int p = 0; p < cnt; ++p) {
          gid[p] = P[cursor + p]; row[p] = rowid[gid[p]] - kBatchRows * b; lr[p] = rk[pv[gid[p]]] - rank0[C] + 1; q[p] = pq[gid[p]];
          B[p] = Bin[(size_t)kBatchRows * b + row[p]];
        }
        // tree query (state before the window)
        Rec t[64]; double gcol[64]; int winj[64];
        for (int p = 0; p < cnt; ++p) {
          Rec best_t{0.0, 0u};
          int x = lr[p];
          for (int l = 0; l < LV; ++l) { best_t = lexmax(best_t, tree[x]); x &= x - 1; }
          t[p] = best_t; gcol[p] = best_t.f; winj[p] = -1;
        }
        // dominance among the window's own points: sequential sweep
        for (int j = 0; j < cnt; ++j) {
          const double fj = q[j] + std::max(gcol[j], B[j].f);
          for (int p = j + 1; p < cnt; ++p)
            if (lr[p] >= lr[j] && fj >= gcol[p]) { gcol[p] = fj; winj[p] = j; }
        }
        Rec me[64];
        for (int p = 0; p < cnt; ++p) {
          const double F = q[p] + std::max(gcol[p], B[p].f);
          const Rec colc{gcol[p], winj[p] >= 0 ? (uint32_t)gid[winj[p]] + 1u : t[p].id1};
          const Rec c = lexmax(colc, B[p]);
          pred[gid[p]] = (int32_t)c.id1 - 1;
          me[p] = Rec{F, (uint32_t)gid[p] + 1u};
        }
        // two-phase tree update: max on the sums, then the ids where the sum is ours
        for (int p = 0; p < cnt; ++p) {
          int x = lr[p];
          for (int l = 0; l < LV && x <= w; ++l) { if (me[p].f > tree[x].f) tree[x].f = me[p].f; x += x & -x; }   // past the column's end: done
        }
        for (int p = 0; p < cnt; ++p) {
          int x = lr[p];
          for (int l = 0; l < LV && x <= w; ++l) { if (tree[x].f == me[p].f && me[p].id1 > tree[x].id1) tree[x].id1 = me[p].id1; x += x & -x; }
        }
        // running maximum per row
        Rec run = M;
        for (int p = 0; p < cnt; ++p) {
          run = lexmax(run, me[p]);
          if (p == cnt - 1 || row[p + 1] != row[p]) rowmax[row[p]] = lexmax(rowmax[row[p]], run);
        }
        M = run;
        cursor += (size_t)cnt;
        if (cnt < 64) break;
      }
      Rec fill = Mstart;
      for (int r = 0; r < kBatchRows; ++r) {
        fill = lexmax(fill, rowmax[r]);
        Bout[(size_t)kBatchRows * b + r] = lexmax(Bin[(size_t)kBatchRows * b + r], fill);
      }
    }
    Bin.swap(Bout);
  }
  *best = (int64_t)Bin[(size_t)n_batches * kBatchRows - 1].id1 - 1;
  return 0;
}
