#!/bin/bash
# Round-4 evidence, run on the GPU box from the repo root: bash profiles/tools/collect_r04.sh
# (writes under gpurun_out/r04/; the summaries to keep are copied into profiles/ by hand)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04; mkdir -p $O; cd $R
# 1. host LP worker-count sweep at configs[2]
for w in 16 24 32; do
  timeout 600 python bench.py --steps 20 --warmup 5 --pipeline $w --no-secondary --no-pcie --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(json.dumps(dict(workers=r['pipeline']['lp_worker_processes'], value=r['value'], lp_s_per_solve=r['host_s_per_step']['lp'], lp_solves_per_s=r['lp_solves_per_s_host'], gpu_stage_pairs_per_s=r['gpu_stage_pairs_per_s'], gemm_ms=r['stage_ms_per_step']['gemm_ms'], frac=r['roofline']['frac'])))" >> $O/worker_sweep_cfg2.jsonl
done
# 2. PMC passes on the configs[2] pair (stereo): MFMA / issue counters, FETCH_SIZE, WRITE_SIZE
bash profiles/tools/pmc_match.sh bf16 7200 r04cfg2 2 > $O/pmc_cfg2.json 2> $O/pmc_cfg2.err
# 3. kernel trace + stats of the bench command itself
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_cfg2 -- python3 $R/bench.py --steps 20 --warmup 5 --no-secondary --no-pcie --no-cpu-baseline > $O/prof_cfg2.log 2>&1
cd $R
cat $O/worker_sweep_cfg2.jsonl; cat $O/pmc_cfg2.json; ls $O/prof_cfg2/*/ | head; tail -2 $O/prof_cfg2.log | cut -c1-600
