#!/bin/bash
# kernel trace + stats of one workload's batch: bash profiles/tools/r05_trace_cfg.sh <cfg1|cfg2|cfg3> <steps> [tag]
WL=$1; STEPS=$2; TAG=${3:-$WL}
R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
mkdir -p $R/gpurun_out/r05
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r05/trace_$TAG -- python3 $R/bench.py --workload $WL --steps $STEPS --warmup 8 --no-cpu-baseline --no-pcie --no-secondary > $R/gpurun_out/r05/trace_$TAG.json 2> $R/gpurun_out/r05/trace_$TAG.err
cd $R
T=$(ls gpurun_out/r05/trace_$TAG/*/*kernel_trace.csv | head -1)
S=$(ls gpurun_out/r05/trace_$TAG/*/*kernel_stats.csv | head -1)
python3 profiles/tools/gemm_gap_trace.py $T > gpurun_out/r05/trace_${TAG}_gaps.json
cp $S gpurun_out/r05/trace_${TAG}_kernel_stats.csv
rm -rf gpurun_out/r05/trace_$TAG            # the raw trace is hundreds of MB
cat gpurun_out/r05/trace_${TAG}_gaps.json; head -12 gpurun_out/r05/trace_${TAG}_kernel_stats.csv | cut -c1-200
