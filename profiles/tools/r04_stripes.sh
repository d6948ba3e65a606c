#!/bin/bash
# stripe length sweep of k_match_bf16 (2 h mono pair; kernel time of the launch, best of 3)
R=$GRAFT_REPO_ROOT; cd $R
for t in 48 96 192 384 768 1536 7000; do
  echo "== stripe tiles $t"
  DALIGN_BF16_STRIPE_TILES=$t timeout 300 python tests/gpu_bench_match.py 7200 1 bf16 2>&1 | cut -c1-250
done
