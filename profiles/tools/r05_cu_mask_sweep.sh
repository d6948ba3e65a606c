#!/bin/bash
# CU-mask sweep (round 5): the chain DP's streams confined to k CUs per XCD (or N whole XCDs), the main stream unmasked or
# given the complement.  Per setting: the configs[1] batch (GPU-bound: the DP of pair k beside the GEMM of pair k + 1) and the
# 2 h pair's chain bench (GEMM alone / beside the DP, DP time).   -> gpurun_out/r05/cu_mask_sweep.jsonl
set -u
mkdir -p gpurun_out/r05
OUT=gpurun_out/r05/cu_mask_sweep.jsonl
: > $OUT
run() {
  local chain="$1" main="$2"
  export DALIGN_CHAIN_CUS="$chain" DALIGN_MAIN_CUS="$main"
  [ -z "$chain" ] && unset DALIGN_CHAIN_CUS
  [ -z "$main" ] && unset DALIGN_MAIN_CUS
  echo "{\"setting\": {\"chain_cus\": \"$chain\", \"main_cus\": \"$main\"}}" >> $OUT
  timeout 600 python3 bench.py --workload cfg1 --steps 96 --warmup 8 --no-cpu-baseline --no-pcie --no-secondary 2>/dev/null \
    | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print(json.dumps({"cfg1": {k: d[k] for k in ("value","ms_per_step","measured_pairs_per_s","gpu_stage_pairs_per_s","bound","stage_ms_per_step","lp_worker_utilisation","lp_solves_per_s_rank")}, "frac": d["roofline"]["frac"], "gpu_thread": d["host_s_per_step"]["gpu_thread"], "intervals": d["host_s_per_step"]["intervals"]}))' >> $OUT
  [ -n "${SKIP_CHAIN:-}" ] || timeout 600 python3 tests/gpu_bench_chain.py 7200 2>/dev/null \
    | python3 -c 'import json,sys; d=json.loads(sys.stdin.readline()); print(json.dumps({"chain_2h": {k: d[k] for k in ("device_chain_ms","columns","gemm_ms_alone","gemm_ms_beside_chain","overlap","identical_to_host")}}))' >> $OUT
}
for s in "$@"; do
  chain="${s%%:*}"; main="${s#*:}"; [ "$main" = "$s" ] && main=""
  run "$chain" "$main"
done
cat $OUT
