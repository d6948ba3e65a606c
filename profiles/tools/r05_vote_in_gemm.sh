#!/bin/bash
# Item "vote where the record is born": the shipped library against the ablation build that does a vote's work in the GEMM's
# rare path and hands k_verify a voted list (make variant NAME=fakevote EXTRA="-DDA_DBG_BF_FAKEVOTE -DDA_DBG_VERIFY_ALLPASS"),
# same box, 2 h stereo pair, alternating.   -> gpurun_out/r05/vote_in_gemm.jsonl
mkdir -p gpurun_out/r05
OUT=gpurun_out/r05/vote_in_gemm.jsonl; : > $OUT
for rep in 1 2; do
  python3 tests/gpu_probe_gaps.py cfg2 8 >> $OUT 2>/dev/null
  DALIGN_LIB=$PWD/describealign_amd/libdalign_fakevote.so python3 tests/gpu_probe_gaps.py cfg2 8 >> $OUT 2>/dev/null
done
cat $OUT
