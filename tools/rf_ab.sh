cd ${GRAFT_REPO_ROOT:-/root/repo}; export TMPDIR=/tmp
cp selfrec_amd/lib/libselfrec_hip.so /tmp/orig.so
for N in ${RF_LIBS:-base rf1 rf2 rf4 rf7}; do
  [ "$N" = base ] || cp tools/spmm_lab/alt/libselfrec_hip_$N.so selfrec_amd/lib/libselfrec_hip.so
  rm -rf gpurun_out/rfab; (cd /tmp && LOSS_PROBE_ITERS=100 LOSS_PROBE_PRECISION=f32 timeout 300 rocprofv3 --kernel-trace --stats -d $OLDPWD/gpurun_out/rfab -o trace -- python $OLDPWD/tools/loss_probe.py > $OLDPWD/gpurun_out/rfab.log 2>&1)
  f=$(find gpurun_out/rfab -name "*.db" | head -1); echo "== $N"; python tools/rocpd_stats.py "$f" | grep -E "rows_finish|nce_finish_bpr2"
  cp /tmp/orig.so selfrec_amd/lib/libselfrec_hip.so
done
rm -rf gpurun_out/rfab
