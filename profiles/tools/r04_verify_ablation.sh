cd /tmp; export TMPDIR=/tmp
for lib in libdalign.so libdalign_nostep2.so libdalign_novote.so libdalign_nothing.so; do
  rm -rf /tmp/vp; DALIGN_LIB=$GRAFT_REPO_ROOT/describealign_amd/$lib rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/vp -- python3 $GRAFT_REPO_ROOT/tests/gpu_pmc_target.py bf16 7200 2 > /dev/null 2>&1
  echo $lib $(grep "k_verify" /tmp/vp/*/*kernel_stats.csv | cut -d, -f2-4)
done
