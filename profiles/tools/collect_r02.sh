#!/bin/bash
# Round-2 evidence, run on the GPU box from the repo root: bash profiles/tools/collect_r02.sh
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
python bench.py --steps 20 --warmup 5 > $O/r02_bench_cfg2_bf16.json 2> $O/r02_bench.err
python bench.py --steps 128 --warmup 5 --no-secondary --no-pcie --no-cpu-baseline > $O/r02_bench_cfg2_bf16_128steps.json 2>> $O/r02_bench.err
python bench.py --workload cfg1 --steps 256 --warmup 8 --no-pcie --no-cpu-baseline > $O/r02_bench_cfg1_f32_256steps.json 2>> $O/r02_bench.err
python tests/gpu_bench_chain.py 1320 7200 > $O/r02_chain_bench.jsonl 2>> $O/r02_bench.err
python tests/gpu_bench_match.py 7200 1 > $O/r02_match_kernels_7200.jsonl 2>> $O/r02_bench.err
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/r02_prof_cfg2 -- python3 $R/bench.py --steps 20 --warmup 5 --no-secondary --no-pcie --no-cpu-baseline > $O/r02_prof_cfg2.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/r02_prof_cfg1 -- python3 $R/bench.py --workload cfg1 --steps 64 --warmup 8 --no-pcie --no-cpu-baseline > $O/r02_prof_cfg1.log 2>&1
cd $R
bash profiles/tools/pmc_match.sh bf16 7200 r02cfg2 2 > $O/r02_pmc_bf16_cfg2.json 2>> $O/r02_bench.err
ls $O/r02_prof_cfg2/*/ $O/r02_prof_cfg1/*/ 2>/dev/null | head
