#!/bin/bash
# Round-6 evidence.  This script is the last step; the other files of profiles/r06_* came from these commands (each run on a lease box
# from the repo root, outputs under gpurun_out/):
#   python profiles/tools/host_cpu_quota_probe.py                         -> profiles/r06_host_cpu_quota_probe.txt   (cgroup cpu.max = 16 CPUs)
#   python profiles/tools/lp_host_scaling.py tests/golden/lp/e7200s.npz 20 -> profiles/r06_lp_host_scaling.jsonl      (LP throughput vs workers, by LP size)
#   python tests/gpu_dump_fit_points.py gpurun_out/lp <cases>             -> tests/golden/lp/*.npz                   (the LP instances of the synthetic pairs)
#   build container: gpurun_out/scratch experiments (windows, stitched bases, merge trees, dual form, IPM, scalings)
#                                                                         -> profiles/r06_lp_decomposition.txt
#   python tests/gpu_stress_lp_tree.py 60 [seed]                          -> profiles/r06_stress_lp_tree*.jsonl      (tree vs reference call through pass 2)
#   DALIGN_DIST_BACKEND=gloo python tests/gpu_tiled_long_pair.py 28800 8  -> profiles/r06_tiled_8h_8ranks.json
#   python bench.py                                                       -> profiles/r06_bench_default*.json        (five runs on different lease boxes)
#   python bench.py --workload cfg2 --steps 24 ... --gpu-streams 1|2|3    -> DESIGN.md 9.6 (two feeding contexts: +4 %)
# This script, one call on the GPU box: kernel trace + stats of the configs[2] batch (the headline's command with the auxiliary
# measurements off), the FETCH_SIZE / WRITE_SIZE passes of the similarity GEMM, then the driver's command (python3 bench.py, no flags)
set -u
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r06
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r06/trace_cfg2 -- python3 $R/bench.py --workload cfg2 --steps 24 --warmup 8 --no-cpu-baseline --no-pcie --no-secondary > $R/gpurun_out/r06/trace_cfg2.json 2> $R/gpurun_out/r06/trace_cfg2.err
cd $R
S=$(ls gpurun_out/r06/trace_cfg2/*/*kernel_stats.csv | head -1)
T=$(ls gpurun_out/r06/trace_cfg2/*/*kernel_trace.csv | head -1)
cp $S gpurun_out/r06/bench_cfg2_bf16_kernel_stats.csv
python3 profiles/tools/gemm_gap_trace.py $T > gpurun_out/r06/trace_cfg2_gaps.json 2>/dev/null
rm -rf gpurun_out/r06/trace_cfg2
head -8 gpurun_out/r06/bench_cfg2_bf16_kernel_stats.csv | cut -c1-180
bash profiles/tools/pmc_match.sh bf16 7200 r06cfg2 2 > gpurun_out/r06/pmc_cfg2_bf16.json 2> gpurun_out/r06/pmc_cfg2_bf16.err
cat gpurun_out/r06/pmc_cfg2_bf16.json
rm -rf gpurun_out/pmc_r06cfg2_*
t0=$(date +%s)
python3 bench.py > gpurun_out/r06/bench_default.json 2> gpurun_out/r06/bench_default.err
echo "driver command: exit $? in $(( $(date +%s) - t0 )) s"
python3 - <<'PY'
import json
d = json.load(open("gpurun_out/r06/bench_default.json"))
print({k: d.get(k) for k in ("value", "ms_per_step", "steps", "bound", "gpu_stage_pairs_per_s", "lp_solves_per_s_rank", "measured_pairs_per_s", "single_pair_latency_s", "lp_method")})
print("roofline", {k: d["roofline"].get(k) for k in ("frac", "frac_all_launches", "avg_launch_ms", "alone_launch_ms_same_device")})
print("stage", d["stage_ms_per_step"]); print("host_lp", {k: v for k, v in d.get("host_lp", {}).items() if k != "note"})
for s in ("secondary", "secondary_cfg3"):
  x = d[s]; print(s, round(x["value"], 3), x["bound"], x["stage_ms_per_step"], x.get("lp_method", {}).get("timed_pairs"))
for s in ("finite_batch_cfg3", "finite_batch_cfg2"):
  x = d[s]; print(s, {k: x.get(k) for k in ("pairs", "wall_s", "value", "mean_lp_s", "max_offset_err_vs_injected_ms", "error")})
print("cpu", {k: d["cpu_baseline"].get(k) for k in ("value", "sample_value", "seconds", "max_offset_err_vs_cpu_ms")})
PY
