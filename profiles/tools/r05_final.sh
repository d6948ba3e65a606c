#!/bin/bash
# final evidence of the round, one call: kernel trace + stats of the configs[2] and configs[1] batches, then the driver's command x3
set -u
bash profiles/tools/r05_trace_cfg.sh cfg2 24 > gpurun_out/r05_final_trace_cfg2.txt 2>&1
bash profiles/tools/r05_trace_cfg.sh cfg1 96 > gpurun_out/r05_final_trace_cfg1.txt 2>&1
bash profiles/tools/r05_bench_default.sh 3 2>&1 | grep -v "^feature\|^cpu\|^pcie"
head -4 gpurun_out/r05/trace_cfg2_kernel_stats.csv | cut -c1-160
