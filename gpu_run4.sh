cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 600 python tools/gemm_bench.py > gpurun_out/gemm_bench4.log 2>&1; echo "gemm rc=$?"; cat gpurun_out/gemm_bench4.log | tail -14
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu4.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/pytest_gpu4.log
timeout 600 python bench.py --steps 8 --warmup 2 --no-cpu-baseline > gpurun_out/bench4_b32.log 2>&1; tail -1 gpurun_out/bench4_b32.log | cut -c1-900
