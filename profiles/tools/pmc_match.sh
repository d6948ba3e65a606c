#!/bin/bash
# rocprofv3 PMC passes over one feature + match pass (tests/gpu_pmc_target.py).  Run on the GPU box:
#   bash profiles/tools/pmc_match.sh <f32|bf16> <seconds> <tag> [channels] [lib]
# Separate passes per counter set, --kernel-trace only (no other trace domains).  Prints one JSON object
# with the per-launch means of the similarity-GEMM kernel, summed over the 8 XCDs.
PREC=$1; SECS=$2; TAG=$3; CH=${4:-1}; LIB=$5
R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
[ -n "$LIB" ] && export DALIGN_LIB=$LIB
for set in "GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_INSTS_SALU" "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD" "FETCH_SIZE" "WRITE_SIZE"; do
  n=$(echo $set | tr ' ' '_' | cut -c1-40)
  timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $R/gpurun_out/pmc_${TAG}_$n -- python3 $R/tests/gpu_pmc_target.py $PREC $SECS $CH > $R/gpurun_out/pmc_${TAG}_$n.log 2>&1
done
cd $R
python3 - "$PREC" "$TAG" <<'PY'
import csv, glob, collections, json, sys
prec, tag = sys.argv[1], sys.argv[2]
out = {}
for f in sorted(glob.glob(f"gpurun_out/pmc_{tag}_*/*/*counter_collection.csv")):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if ("k_match_" + prec) in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        out[k] = sum(v) / len(v)
print(json.dumps(out))
PY
