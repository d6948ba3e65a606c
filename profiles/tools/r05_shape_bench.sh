#!/bin/bash
# MFMA shape micro-benchmark (profiles/tools/microbench/shape_bench.hip) on all CUs and on one CU.
# usage on the GPU box, from the repo root: bash profiles/tools/r05_shape_bench.sh  -> gpurun_out/r05/shape_bench.txt
set -u
mkdir -p gpurun_out/r05
B=profiles/tools/microbench/shape_bench
[ -x $B ] || /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 $B.hip -o $B
( $B 256; echo "==== one-block"; $B 1 ) > gpurun_out/r05/shape_bench.txt 2>&1
tail -40 gpurun_out/r05/shape_bench.txt
