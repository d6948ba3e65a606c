#!/bin/bash
# Round-3 evidence, run on the GPU box from the repo root: bash profiles/tools/collect_r03.sh
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
python tests/gpu_bench_chain.py 1320 7200 > $O/r03_chain_bench.jsonl 2> $O/r03_collect.err
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/r03_prof_cfg2 -- python3 $R/bench.py --steps 20 --warmup 5 --no-secondary --no-pcie --no-cpu-baseline > $O/r03_prof_cfg2.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/r03_prof_chain -- python3 $R/tests/gpu_bench_chain.py 7200 > $O/r03_prof_chain.log 2>&1
cd $R
bash profiles/tools/pmc_match.sh bf16 7200 r03cfg2 2 > $O/r03_pmc_bf16_cfg2.json 2>> $O/r03_collect.err
ls $O/r03_prof_cfg2/*/ $O/r03_prof_chain/*/ 2>/dev/null | head
