#!/bin/bash
# L2-miss traffic (FETCH_SIZE, KB) and kernel time of k_match_bf16 against the stripe length (tiles of 32 audio columns per workgroup)
R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
for t in 96 192 384 768 1536 3072; do
  export DALIGN_BF16_STRIPE_TILES=$t
  timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/r04/fetch_$t -- python3 $R/tests/gpu_pmc_target.py bf16 7200 1 > $R/gpurun_out/r04/fetch_$t.log 2>&1
  python3 - $t $R <<'PY'
import csv, glob, sys
t, R = sys.argv[1], sys.argv[2]
vals, dur = [], []
for f in glob.glob(f"{R}/gpurun_out/r04/fetch_{t}/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "k_match_bf16" in r["Kernel_Name"] and r["Counter_Name"] == "FETCH_SIZE":
            vals.append(float(r["Counter_Value"]))
for f in glob.glob(f"{R}/gpurun_out/r04/fetch_{t}/*/*kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        if "k_match_bf16" in r["Kernel_Name"]:
            dur.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-6)
print(f"stripe {t:>5} tiles: FETCH_SIZE {sum(vals) / max(1, len(vals)) / 1e6:8.2f} GB raw (x2 for 16-byte loads), kernel {sum(dur) / max(1, len(dur)):7.2f} ms under the profiler")
PY
done
