"""GPU-box script (not a pytest): stage-2 chain DP on the device vs the host utility.

  python tests/gpu_bench_chain.py [seconds ...]        default: 1320 7200

For each duration: synthetic mono pair -> features -> match (bf16 GEMM for >= 3000 s) -> the device
DP on the resident match list (HIP-event time) and the host utility on the same list copied out
(wall time, one core), paths compared; then the overlap check: the DP of this pair enqueued on its
own stream while the similarity GEMM of the next pair runs on the context's main stream.
Prints one JSON line per duration."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from describealign_amd import _native, synth  # noqa: E402


def main():
  durations = [float(a) for a in sys.argv[1:]] or [1320.0, 7200.0]
  for sec in durations:
    prec = _native.PREC_BF16 if sec >= 3000 else _native.PREC_F32
    ctx = _native.Context(0, prec)
    pair = synth.make_pair(5, sec, n_jumps=10, first_gap=200.0)
    vf = ctx.features(pair.video, 0); af = ctx.features(pair.audio, 1)
    mi, mv, mq = ctx.match(vf, af)
    st = ctx.stats()
    t0 = time.perf_counter(); gi, gv = ctx.chain_resident(); wall_dev = time.perf_counter() - t0
    dev_ms = ctx.stats()["chain_ms"]; n_cols = int(ctx.stats()["chain_columns"]); col_width = int(ctx.stats()["chain_column_width"])
    # the round-2 kernel (one workgroup, four wavefronts per row) on the same list, for the record
    os.environ["DALIGN_CHAIN_KERNEL"] = "rows"
    ctx.match_begin(vf, af); ctx.match_finish()
    ri, rv = ctx.chain_resident(); rows_ms = ctx.stats()["chain_ms"]
    os.environ.pop("DALIGN_CHAIN_KERNEL")
    same_rows = bool(np.array_equal(ri, gi) and np.array_equal(rv, gv))
    t0 = time.perf_counter(); hi, hv = _native.chain_host(mi, mv, mq); host_s = time.perf_counter() - t0
    same = bool(len(gi) == len(hi) and np.array_equal(gi, hi) and np.array_equal(gv, hv))
    rows = int(len(np.unique(mi)))
    # overlap: chain DP of pair A in flight while pair B's GEMM runs
    n = ctx.match_begin(vf, af); ctx.match_finish()
    t0 = time.perf_counter()
    ticket = ctx.chain_begin()
    ctx.match_begin(vf, af); ctx.match_finish()
    gemm_beside = ctx.stats()["gemm_ms"]
    t_match = time.perf_counter() - t0
    pi, pv = ctx.chain_finish(ticket)
    t_both = time.perf_counter() - t0
    ticket2 = ctx.chain_begin(); ctx.chain_finish(ticket2)
    print(json.dumps(dict(seconds=sec, matches=len(mi), rows=rows, matches_per_row=round(len(mi) / max(rows, 1), 2),
                          video_ranks=int(st["gemm_pairs"] / max(1, len(np.unique(mi)))) if False else None,
                          path=len(gi), identical_to_host=same, device_chain_ms=round(dev_ms, 2), columns=n_cols, column_width=col_width,
                          one_workgroup_kernel_ms=round(rows_ms, 2), one_workgroup_kernel_identical=same_rows,
                          device_wall_ms=round(1e3 * wall_dev, 2), host_chain_ms=round(1e3 * host_s, 2),
                          ns_per_match_device=round(1e6 * dev_ms / max(1, len(mi)), 2),
                          us_per_row_device=round(1e3 * dev_ms / max(1, rows), 3),
                          gemm_ms_alone=round(st["gemm_ms"], 2), gemm_ms_beside_chain=round(gemm_beside, 2),
                          overlap=dict(match_wall_ms=round(1e3 * t_match, 1), match_plus_chain_wall_ms=round(1e3 * t_both, 1)),
                          path_again_identical=bool(np.array_equal(pi, gi)))), flush=True)
    ctx.close()


if __name__ == "__main__":
  main()
