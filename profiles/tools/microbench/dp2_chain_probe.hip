// Microbenchmark behind DESIGN.md section 4.7: what a device version of the second DP (describealign.py:946-990) could reach
// at best.  The DP visits its points in (i, j) order; a point's sum needs, before the next point of the same cluster can be
// evaluated, (1) the frontier entry below j (a search in a small sorted list), (2) its cluster's best entry, (3) the cache
// entries of the last three video frames -- one of which the previous point has just written -- and then writes its own entry,
// the cluster's best and, sometimes, the frontier.  Even with every structure in LDS and nothing ever missing, that is a chain
// of dependent LDS round trips per point.  This kernel runs exactly such a chain (one wavefront, all structures in LDS, the
// three cache reads in parallel lanes, no frontier edits, no global memory in the loop) for n points and reports the time per
// point: the floor under any single-wavefront device version.  The host loop it would replace takes ~48 ns per point.
//   hipcc --offload-arch=gfx950 -O3 -o dp2_chain_probe dp2_chain_probe.hip && ./dp2_chain_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

struct Entry { double j; double cum; int cl; int id; };

__global__ __launch_bounds__(64) void k_probe(const double* __restrict__ pj, const double* __restrict__ pq, const int* __restrict__ pcl, int n,
                                              int n_frontier, double* __restrict__ out) {
  __shared__ Entry s_cache[4096];          // sliding window of the per-frame cache (the real one has a slot per video frame)
  __shared__ Entry s_front[64];            // the frontier: sorted by j, sums increasing
  __shared__ Entry s_clbest[64];
  const int lane = threadIdx.x;
  for (int k = lane; k < 4096; k += 64) s_cache[k] = Entry{0.0, -1e30, -1, -1};
  s_front[lane] = Entry{(double)lane * 1e9 / n_frontier, (double)lane, -1, -1};
  s_clbest[lane] = Entry{0.0, -1000.0, lane, -1};
  __syncthreads();
  double carry = 0.0;
  for (int base = 0; base < n; base += 64) {
    // a wavefront's worth of points is fetched at once (the inputs do not depend on the DP)
    const double my_j = base + lane < n ? pj[base + lane] : 0.0;
    const double my_q = base + lane < n ? pq[base + lane] : 0.0;
    const int my_cl = base + lane < n ? pcl[base + lane] : 0;
    const int cnt = n - base < 64 ? n - base : 64;
    for (int t = 0; t < cnt; ++t) {
      const double j = __shfl(my_j, t), q = __shfl(my_q, t);
      const int cl = __shfl(my_cl, t);
      // (1) frontier: every lane compares its entry, the last one not above j wins (bisect_right by ballot)
      const unsigned long long below = __ballot(lane < n_frontier && s_front[lane].j <= j + carry * 0.0);
      const int pos = 63 - __builtin_clzll(below | 1ull);
      double best = s_front[pos].cum - 1000.0;
      // (2) the cluster's best
      const double cb = s_clbest[cl & 63].cum;
      best = cb >= best ? cb : best;
      // (3) the three cache slots, one per lane, then a three-way maximum
      const int jj = (int)j;
      double c = -1e30;
      if (lane < 3) {
        const Entry e = s_cache[(jj - 2 + lane) & 4095];
        const double skew = (j - e.j);
        c = e.cl == cl ? e.cum : e.cum - (100.0 + 100.0 * skew * skew);
      }
      const double c0 = __shfl(c, 0), c1 = __shfl(c, 1), c2 = __shfl(c, 2);
      best = c0 >= best ? c0 : best; best = c1 >= best ? c1 : best; best = c2 >= best ? c2 : best;
      const double cum = best + q;
      if (lane == 0) {
        s_cache[jj & 4095] = Entry{j, cum, cl, base + t};
        if (s_clbest[cl & 63].cum < cum - 50.0) s_clbest[cl & 63] = Entry{j, cum - 50.0, cl, base + t};
      }
      carry = cum;
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    }
  }
  if (lane == 0) out[0] = carry;
}

int main() {
  const int n = 1800000;                                  // points of a 2 h pair
  std::vector<double> j(n), q(n); std::vector<int> cl(n);
  for (int k = 0; k < n; ++k) { j[k] = 0.8 * k + (k % 7) * 0.1; q[k] = 1.0 + (k % 13) * 0.25; cl[k] = (k / 20000) % 40; }
  double *dj, *dq, *dout; int* dcl;
  hipMalloc(&dj, n * 8); hipMalloc(&dq, n * 8); hipMalloc(&dcl, n * 4); hipMalloc(&dout, 8);
  hipMemcpy(dj, j.data(), n * 8, hipMemcpyHostToDevice); hipMemcpy(dq, q.data(), n * 8, hipMemcpyHostToDevice);
  hipMemcpy(dcl, cl.data(), n * 4, hipMemcpyHostToDevice);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(k_probe, dim3(1), dim3(64), 0, 0, dj, dq, dcl, n, 24, dout);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms = 0.f; hipEventElapsedTime(&ms, e0, e1);
    double r = 0; hipMemcpy(&r, dout, 8, hipMemcpyDeviceToHost);
    std::printf("second-DP chain probe: %d points, %.1f ms, %.1f ns per point (checksum %.3f)\n", n, ms, 1e6 * ms / n, r);
  }
  return 0;
}
