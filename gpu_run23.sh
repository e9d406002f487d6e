cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for mode in 1 stats 0; do
  TSII_FUSE_BN=$mode timeout 600 python tools/profile_step.py > gpurun_out/profile_step23_$mode.log 2>&1; echo "== FUSE_BN=$mode"; grep -E "^  tsii_(pw|dw|bn)|total" gpurun_out/profile_step23_$mode.log | cut -c1-60 | head -16
done
