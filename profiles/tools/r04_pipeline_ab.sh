#!/bin/bash
# the batch pipeline at configs[2]: admission pacing + worker-start staggering on / off, same box, 40 timed pairs
R=$GRAFT_REPO_ROOT; cd $R
cat > /tmp/r04_pick.py <<'PY'
import json, sys
r = json.loads(sys.stdin.read().strip().splitlines()[-1])
h = r["host_s_per_step"]
print(json.dumps(dict(value=round(r["value"], 3), measured=r["measured_pairs_per_s"], util=r.get("lp_worker_utilisation"), lp=h["lp"],
                      gemm_ms=round(r["stage_ms_per_step"]["gemm_ms"], 1), gpu_thread=h.get("gpu_thread"), iv=h.get("intervals"))))
PY
run() { timeout 600 python bench.py --steps 40 --warmup 5 --no-secondary --no-pcie --no-cpu-baseline 2>/dev/null | python3 /tmp/r04_pick.py; }
echo "== pace + stagger (default)"; run
echo "== DALIGN_PIPELINE_PACE=0"; DALIGN_PIPELINE_PACE=0 run
