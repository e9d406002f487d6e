cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu24.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/pytest_gpu24.log
timeout 600 python tools/profile_step.py > gpurun_out/profile_step24.log 2>&1; grep -E "^  tsii_|total" gpurun_out/profile_step24.log | cut -c1-60 | head -22; grep -E "^tsii_dense" gpurun_out/profile_step24.log | cut -c1-150
timeout 600 python bench.py --steps 8 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-400
timeout 600 python bench.py --model ImageFillOrigin --batch 32 --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-200
