cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 300 python tools/dw_bench.py 2>&1 | grep "^dw"
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu12.log 2>&1; echo "pytest rc=$?"; tail -2 gpurun_out/pytest_gpu12.log
timeout 600 python bench.py --steps 8 --warmup 2 --no-cpu-baseline > gpurun_out/bench12_b32.log 2>&1; tail -1 gpurun_out/bench12_b32.log | cut -c1-330
