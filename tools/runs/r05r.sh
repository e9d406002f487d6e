#!/bin/bash
# round 5, run r: is the ~1.1 PF/s of the 256 x 256 kernels the power cap?  same kernel, operands normal / uniform / zeros, 4096^3 .. 65536 x 4096^2, 1 and 10 launches
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R"; mkdir -p gpurun_out/r05r
for fill in normal uniform zeros; do
  timeout 200 python tools/bf16_bench.py --gemm "4096,4096,4096;8192,8192,8192;65536,4096,4096" --fill $fill --iters 10 2>&1 | grep -v amdgpu.ids
  timeout 200 python tools/bf16_bench.py --gemm "4096,4096,4096;65536,4096,4096" --fill $fill --iters 1 2>&1 | grep -v amdgpu.ids
done > gpurun_out/r05r/gemm_fill.log 2>&1
cat gpurun_out/r05r/gemm_fill.log
