#!/bin/bash
# round 3, run 2: wave priorities of the producer / consumer GEMM (TSII_GEMM_PC_OPT: bits 0-1 consumers, 2-3 producers)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; mkdir -p gpurun_out
for o in 1 0 12 4 13; do
  TSII_GEMM_PC_OPT=$o timeout 300 python tools/gemm_bench.py --only nt --iters 5 --shapes 1,4,6,8,9,11,13 > gpurun_out/r03b_gemm_opt$o.log 2>&1; echo "opt $o rc=$?"
  grep -v amdgpu.ids gpurun_out/r03b_gemm_opt$o.log
done
