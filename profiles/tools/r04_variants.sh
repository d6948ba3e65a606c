#!/bin/bash
# ablation builds of k_match_bf16 on the 2 h mono pair (kernel time of the launch, best of 3)
R=$GRAFT_REPO_ROOT; cd $R
for v in "" _noepi _noemit _noload _noepiemit _bare; do
  lib=$R/describealign_amd/libdalign$v.so
  echo "== ${v:-base}"
  DALIGN_LIB=$lib timeout 300 python tests/gpu_bench_match.py 7200 1 bf16 2>&1 | cut -c1-200
done
