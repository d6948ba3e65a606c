#!/usr/bin/env python3
"""One line per variant of gpurun_out/r05/cfg1_variants.jsonl (profiles/tools/r05_cfg1_variants.sh)."""
import json, sys
for l in open(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/r05/cfg1_variants.jsonl"):
  d = json.loads(l)
  print(d["variant"], "value", round(d["value"], 3), "ms/step", round(d["ms_per_step"], 2), "bound", d["bound"], "util", d["util"], "lp_rate", d["lp_rate"],
        "stage_wall_ms", round(1e3 * d["host"]["gpu_match_stage_wall"], 1), "lp_s", d["host"]["lp"], "refine_s", d["host"]["refine"], "frac", round(d["frac"], 4))
