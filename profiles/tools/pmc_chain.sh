#!/bin/bash
# rocprofv3 PMC passes over one match + chain DP (tests/gpu_pmc_target.py ... chain).  Run on the GPU box:
#   bash profiles/tools/pmc_chain.sh <seconds> <tag>
# Separate passes per counter set, --kernel-trace only.  Prints one JSON object with the counters of
# k_chain_columns and k_chain_backtrack (one launch each), summed over the 8 XCDs.
SECS=$1; TAG=$2
R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
for set in "GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_SALU" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_WR" "FETCH_SIZE" "WRITE_SIZE"; do
  n=$(echo $set | tr ' ' '_' | cut -c1-40)
  timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $R/gpurun_out/pmcc_${TAG}_$n -- python3 $R/tests/gpu_pmc_target.py bf16 $SECS 1 chain > $R/gpurun_out/pmcc_${TAG}_$n.log 2>&1
done
cd $R
python3 - "$TAG" <<'PY'
import csv, glob, collections, json, sys
tag = sys.argv[1]
out = {}
for f in sorted(glob.glob(f"gpurun_out/pmcc_{tag}_*/*/*counter_collection.csv")):
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(f)):
        for k in ("k_chain_columns", "k_chain_backtrack", "k_col_gather"):
            if k in r["Kernel_Name"]:
                agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    for k, d in agg.items():
        out.setdefault(k, {}).update(d)
print(json.dumps(out))
PY
