#!/bin/bash
# all PMC passes of the round in one call -> gpurun_out/pmc_{cfg2,cfg3,cfg1}.json (+ the launch time of the GRBM pass)
for spec in "bf16 7200 cfg2 2" "bf16 1800 cfg3 1" "f32 1320 cfg1 1"; do
  set -- $spec
  bash profiles/tools/pmc_match.sh $1 $2 $3 $4 > gpurun_out/pmc_$3.json 2> gpurun_out/pmc_$3.err
  python3 - $1 $3 <<'PY'
import csv, glob, json, sys
prec, tag = sys.argv[1], sys.argv[2]
f = glob.glob(f"gpurun_out/pmc_{tag}_GRBM*/*/*kernel_trace.csv")[0]
d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-6 for r in csv.DictReader(open(f)) if ("k_match_" + prec) in r["Kernel_Name"]]
pm = json.load(open(f"gpurun_out/pmc_{tag}.json"))
pm["launch_ms_in_grbm_pass"] = sum(d) / len(d)
json.dump(pm, open(f"gpurun_out/pmc_{tag}.json", "w"))
print(tag, json.dumps(pm))
PY
done
