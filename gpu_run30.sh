cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu30.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/pytest_gpu30.log
timeout 600 python tools/profile_step.py > gpurun_out/profile_step30.log 2>&1; grep -E "^  tsii_|total" gpurun_out/profile_step30.log | cut -c1-60 | head -18; grep -E "^tsii_dw" gpurun_out/profile_step30.log | cut -c1-120 | head -12
