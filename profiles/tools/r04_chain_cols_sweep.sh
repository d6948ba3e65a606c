mkdir -p gpurun_out/r04
for nc in 128 192 256 320 384 512 768; do
  echo "== cols $nc"
  DALIGN_CHAIN_COLS=$nc DALIGN_LIB=$PWD/describealign_amd/libdalign_dbg.so timeout 300 python profiles/tools/chain_timeline.py 7200 gpurun_out/r04/tl_$nc.txt 2>&1 | grep -v amdgpu.ids | grep "^matches\|^sample 30\|^sample 0\|stamps" | tail -3
done
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r04/chain_trace -- python3 $GRAFT_REPO_ROOT/tests/gpu_pmc_target.py bf16 7200 2 chain > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
f=$(ls gpurun_out/r04/chain_trace/*/*kernel_stats.csv | head -1); head -30 $f | cut -c1-150
