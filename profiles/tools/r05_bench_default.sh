#!/bin/bash
# the driver's command (python3 bench.py, no flags), N times back to back -> gpurun_out/r05/bench_default_<k>.json + a summary
N=${1:-1}
mkdir -p gpurun_out/r05
for k in $(seq 1 $N); do
  t0=$(date +%s)
  python3 bench.py > gpurun_out/r05/bench_default_$k.json 2> gpurun_out/r05/bench_default_$k.err
  echo "run $k: exit $? in $(( $(date +%s) - t0 )) s"
  python3 - gpurun_out/r05/bench_default_$k.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print({k: d[k] for k in ("value", "ms_per_step", "steps", "bound", "n_gpus", "ranks_in_group", "gpu_stage_pairs_per_s", "measured_pairs_per_s", "lp_solves_per_s_rank", "single_pair_latency_s")})
print("roofline", {k: d["roofline"].get(k) for k in ("frac", "frac_all_launches", "avg_launch_ms", "avg_launch_ms_all_launches", "alone_launch_ms_same_device", "alone_frac_same_device", "traffic_over_algorithmic")})
print("stage", d["stage_ms_per_step"]); print("host_lp", {k: v for k, v in d.get("host_lp", {}).items() if k != "note"})
for s in ("secondary", "secondary_cfg3"):
  x = d[s]; print(s, round(x["value"], 3), round(x["ms_per_step"], 2), x["steps"], x["bound"], x["timed_region"]["timed_s"], round(x["roofline"]["frac"], 4), x["roofline"].get("frac_all_launches"), x["roofline"].get("traffic"), x["stage_ms_per_step"])
print("feature_stage", d["feature_stage"]); print("cpu", d["cpu_baseline"]["value"], d["cpu_baseline"]["max_offset_err_vs_cpu_ms"]); print("pcie", d["pcie_inclusive"]["value"])
PY
done
