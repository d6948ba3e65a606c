#!/bin/bash
# Round-4 closing evidence, run on the GPU box from the repo root: bash profiles/tools/collect_r04_final.sh
#  1. the default bench line                       -> gpurun_out/r04/bench_default.json
#  2. rocprofv3 --kernel-trace --stats of the same command (default flags) -> gpurun_out/r04/prof_default/
#  3. PMC passes of the chain DP on the 2 h pair   -> gpurun_out/r04/pmc_chain.json
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04; mkdir -p $O; cd $R
timeout 900 python3 bench.py > $O/bench_default.json 2> $O/bench_default.err
cd /tmp; export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_default -- python3 $R/bench.py --no-secondary --no-pcie --no-cpu-baseline > $O/prof_default.log 2>&1
cd $R
bash profiles/tools/pmc_chain.sh 7200 r04 > $O/pmc_chain.json 2> $O/pmc_chain.err
tail -c 600 $O/bench_default.json; echo; f=$(ls $O/prof_default/*/*kernel_stats.csv | head -1); head -12 $f | cut -c1-160; tail -c 600 $O/pmc_chain.json
