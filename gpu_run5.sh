cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu5.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/pytest_gpu5.log
timeout 600 python tools/profile_step.py > gpurun_out/profile_step5.log 2>&1; echo "profile rc=$?"; head -18 gpurun_out/profile_step5.log | tail -17
grep -E "tsii_dw_|tsii_dense" gpurun_out/profile_step5.log | head -24 | cut -c1-140
timeout 600 python bench.py --steps 8 --warmup 2 --no-cpu-baseline > gpurun_out/bench5_b32.log 2>&1; tail -1 gpurun_out/bench5_b32.log | cut -c1-330
