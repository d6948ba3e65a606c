#!/bin/bash
# configs[1] batch (GPU-bound) against the number of GPU-feeding contexts and the chain DP's CU mask -> gpurun_out/r05/gpu_streams.jsonl
set -u
mkdir -p gpurun_out/r05
OUT=gpurun_out/r05/gpu_streams.jsonl
: > $OUT
for s in "$@"; do
  streams="${s%%:*}"; chain="${s#*:}"
  if [ -n "$chain" ]; then export DALIGN_CHAIN_CUS="$chain"; else unset DALIGN_CHAIN_CUS; fi
  timeout 600 python3 bench.py --workload cfg1 --steps 160 --warmup 8 --gpu-streams $streams --no-cpu-baseline --no-pcie --no-secondary 2>/dev/null \
    | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print(json.dumps({"gpu_streams": '$streams', "chain_cus": "'$chain'", "value": d["value"], "ms_per_step": d["ms_per_step"], "frac": d["roofline"]["frac"], "frac_all": d["roofline"]["frac_all_launches"], "stage_ms": d["stage_ms_per_step"], "gpu_thread": d["host_s_per_step"]["gpu_thread"], "util": d.get("lp_worker_utilisation"), "host_lp": d.get("host_lp")}))' >> $OUT
done
cat $OUT
