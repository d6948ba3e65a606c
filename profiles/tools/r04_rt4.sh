#!/bin/bash
# six resident row tiles (216 AGPRs, 428 registers: no other wave fits beside a GEMM wave) against four (144 AGPRs, 356
# registers: a chain-DP wave of 152 fits on the same SIMD): kernel alone, and inside the bench pipeline
R=$GRAFT_REPO_ROOT; cd $R
for v in "" _rt4; do
  lib=$R/describealign_amd/libdalign$v.so
  echo "== ${v:-six row tiles}"
  DALIGN_LIB=$lib timeout 300 python tests/gpu_bench_match.py 7200 2 bf16 2>&1 | cut -c1-200
  DALIGN_LIB=$lib timeout 600 python bench.py --steps 20 --warmup 5 --no-secondary --no-pcie --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(json.dumps(dict(value=r['value'], gemm_ms=r['stage_ms_per_step']['gemm_ms'], frac=r['roofline']['frac'], chain_ms=r['stage_ms_per_step']['chain_ms'], verify_ms=r['stage_ms_per_step']['verify_ms'], lp=r['host_s_per_step']['lp'])))"
done
