#!/bin/bash
# round-2 GPU run 3: accuracy of the arithmetic modes, the noise-limited seg-net test per mode, GEMM microbench, bench line
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_parity_r2.py -m gpu -q -k "accuracy" > gpurun_out/r02c_accuracy.log 2>&1; echo "accuracy rc=$?"; grep "gemm accuracy" gpurun_out/r02c_accuracy.log
for m in 0 6 8; do
TSII_GEMM_PRODUCTS=$m timeout 900 python -m pytest tests/test_parity_r2.py -m gpu -q -k "seg_nets_256" > gpurun_out/r02c_seg256_mode$m.log 2>&1; echo "seg256 mode $m rc=$?"; grep -E "rel err|passed|failed" gpurun_out/r02c_seg256_mode$m.log | tail -3
done
timeout 1500 python -m pytest tests -m gpu -q --deselect tests/test_parity_r2.py::test_seg_nets_256_vs_oracle_gpu > gpurun_out/r02c_gputests.log 2>&1; echo "gpu tests rc=$?"; tail -3 gpurun_out/r02c_gputests.log
timeout 600 python tools/gemm_bench.py --iters 5 --modes 6,8 > gpurun_out/r02c_gemm.log 2>&1; echo "gemm rc=$?"
cut -c1-200 gpurun_out/r02c_gemm.log
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/r02c_bench.log 2>&1; echo "bench rc=$?"; tail -1 gpurun_out/r02c_bench.log | cut -c1-3000
