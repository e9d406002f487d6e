cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 600 python tools/profile_step.py > gpurun_out/profile_step7.log 2>&1; echo "profile rc=$?"; head -14 gpurun_out/profile_step7.log | tail -13
grep -E "tsii_dw_" gpurun_out/profile_step7.log | head -9 | cut -c1-130
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu7.log 2>&1; echo "pytest rc=$?"; tail -2 gpurun_out/pytest_gpu7.log
timeout 600 python bench.py --steps 8 --warmup 2 --no-cpu-baseline > gpurun_out/bench7_b32.log 2>&1; tail -1 gpurun_out/bench7_b32.log | cut -c1-330
