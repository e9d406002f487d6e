cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu36.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/pytest_gpu36.log
for v in 1 0; do echo "FUSE_BN_BWD=$v"; TSII_FUSE_BN_BWD=$v timeout 600 python tools/profile_step.py > gpurun_out/profile_step36_$v.log 2>&1; grep -E "^  tsii_(bn_act_bwd|dw_bwd_dx|pw_bwd_dx)|total" gpurun_out/profile_step36_$v.log | cut -c1-60 | head -6; done
