cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 600 python tools/profile_step.py > gpurun_out/profile_step10.log 2>&1; echo "profile rc=$?"; head -14 gpurun_out/profile_step10.log | tail -13
grep -E "tsii_dw_fwd|tsii_dw_bwd_dx" gpurun_out/profile_step10.log | head -8 | cut -c1-130
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu10.log 2>&1; echo "pytest rc=$?"; tail -2 gpurun_out/pytest_gpu10.log
timeout 600 python bench.py --steps 8 --warmup 2 --no-cpu-baseline > gpurun_out/bench10_b32.log 2>&1; tail -1 gpurun_out/bench10_b32.log | cut -c1-330
