#!/bin/bash
# configs[1] / configs[3] batches against the number of LP worker processes (chain DP on 8 CUs per XCD) -> gpurun_out/r05/workers_short_pairs.jsonl
set -u
mkdir -p gpurun_out/r05
OUT=gpurun_out/r05/workers_short_pairs.jsonl
: > $OUT
export DALIGN_CHAIN_CUS=8
for s in "$@"; do
  wl="${s%%:*}"; w="${s#*:}"
  timeout 600 python3 bench.py --workload $wl --steps 192 --warmup 8 --pipeline $w --no-cpu-baseline --no-pcie --no-secondary 2>/dev/null \
    | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print(json.dumps({"workload": "'$wl'", "workers": '$w', "value": d["value"], "ms_per_step": d["ms_per_step"], "bound": d["bound"], "gpu_stage_pairs_per_s": d["gpu_stage_pairs_per_s"], "measured_pairs_per_s": d["measured_pairs_per_s"], "lp_solves_per_s_rank": d["lp_solves_per_s_rank"], "frac": d["roofline"]["frac"], "util": d.get("lp_worker_utilisation"), "lp_under_load": d["host_s_per_step"]["lp"], "refine": d["host_s_per_step"]["refine"], "gemm_ms": d["stage_ms_per_step"]["gemm_ms"]}))' >> $OUT
done
cat $OUT
