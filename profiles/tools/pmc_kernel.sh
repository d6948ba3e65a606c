#!/bin/bash
# rocprofv3 PMC passes over tests/gpu_pmc_target.py, reporting the per-launch counter means of kernels whose name
# contains <pattern> (summed over the 8 XCDs per launch).  Run on the GPU box:
#   bash profiles/tools/pmc_kernel.sh <pattern> <f32|bf16> <seconds> <channels> <tag>
PAT=$1; PREC=$2; SECS=$3; CH=$4; TAG=$5
R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
for set in "GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_SALU" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS" "FETCH_SIZE" "WRITE_SIZE"; do
  n=$(echo $set | tr ' ' '_' | cut -c1-40)
  timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $R/gpurun_out/pmck_${TAG}_$n -- python3 $R/tests/gpu_pmc_target.py $PREC $SECS $CH > $R/gpurun_out/pmck_${TAG}_$n.log 2>&1
done
cd $R
python3 - "$PAT" "$TAG" <<'PY'
import csv, glob, collections, json, sys
pat, tag = sys.argv[1], sys.argv[2]
out = {}
for f in sorted(glob.glob(f"gpurun_out/pmck_{tag}_*/*/*counter_collection.csv")):
    per = collections.defaultdict(lambda: collections.defaultdict(float))     # (kernel, dispatch) -> counter -> sum over XCDs
    for r in csv.DictReader(open(f)):
        if pat in r["Kernel_Name"]:
            per[(r["Kernel_Name"].split("(")[0], r["Dispatch_Id"])][r["Counter_Name"]] += float(r["Counter_Value"])
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for (k, _), d in per.items():
        for c, v in d.items(): agg[k][c].append(v)
    for k, d in agg.items():
        for c, v in d.items(): out.setdefault(k, {})[c] = v      # list per launch, in dispatch order
print(json.dumps(out))
PY
