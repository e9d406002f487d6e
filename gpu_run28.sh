cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu28.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/pytest_gpu28.log
timeout 600 python tools/profile_step.py > gpurun_out/profile_step28.log 2>&1; grep -E "^  tsii_|total" gpurun_out/profile_step28.log | cut -c1-60 | head -16; grep -E "^tsii_(dw|pw_bwd_dw )" gpurun_out/profile_step28.log | cut -c1-130 | head -14
