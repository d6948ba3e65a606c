#!/bin/bash
# configs[1] batch under named variants of the pipeline: "name:ENV=VAL,ENV=VAL:extra bench flags"  -> gpurun_out/r05/cfg1_variants.jsonl
mkdir -p gpurun_out/r05
OUT=gpurun_out/r05/cfg1_variants.jsonl; : > $OUT
WL=${WL:-cfg1}; STEPS=${STEPS:-160}
for spec in "$@"; do
  name="${spec%%:*}"; rest="${spec#*:}"; envs="${rest%%:*}"; flags="${rest#*:}"; [ "$flags" = "$rest" ] && flags=""
  ( IFS=','; for kv in $envs; do [ -n "$kv" ] && export "$kv"; done; unset IFS
    python3 bench.py --workload $WL --steps $STEPS --warmup 8 --no-cpu-baseline --no-pcie --no-secondary $flags 2>/dev/null \
    | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print(json.dumps({"variant": "'"$name"'", "value": d["value"], "ms_per_step": d["ms_per_step"], "bound": d["bound"], "frac": d["roofline"]["frac"], "frac_all": d["roofline"]["frac_all_launches"], "alone_ms": d["roofline"].get("alone_launch_ms_same_device"), "host": d["host_s_per_step"], "stage": d["stage_ms_per_step"], "util": d.get("lp_worker_utilisation"), "lp_rate": d["lp_solves_per_s_rank"], "gpu_rate": d["gpu_stage_pairs_per_s"], "host_lp": {k: v for k, v in d.get("host_lp", {}).items() if k != "note"}}))' ) >> $OUT
done
cat $OUT
