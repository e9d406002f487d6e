#!/bin/bash
# round-2 GPU run 1: parity suite + GEMM microbench over arithmetic modes + bench line
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r02a_gputests.log 2>&1; echo "gpu tests rc=$?"; tail -3 gpurun_out/r02a_gputests.log
timeout 600 python tools/gemm_bench.py --iters 5 --only fwd --modes 0,6,3 > gpurun_out/r02a_gemm_fwd.log 2>&1; echo "gemm fwd rc=$?"
timeout 600 python tools/gemm_bench.py --iters 5 --only dx --modes 0,6 > gpurun_out/r02a_gemm_dx.log 2>&1; echo "gemm dx rc=$?"
cat gpurun_out/r02a_gemm_fwd.log | cut -c1-150
for m in 6 0; do
TSII_GEMM_PRODUCTS=$m timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r02a_bench_mode$m.log 2>&1; echo "bench mode $m rc=$?"; tail -1 gpurun_out/r02a_bench_mode$m.log | cut -c1-400
done
