#!/bin/bash
# round 3, run 32: does the persistent kernel now also win the one-stage (K = 32) dX + K6c layers?
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; mkdir -p gpurun_out
for k in 64 32; do
  echo "== TSII_GEMM_PC_BNB_MIN_K=$k"
  TSII_GEMM_PC_BNB_MIN_K=$k timeout 300 python tools/gemm_bench.py --only nt --iters 5 2>&1 | grep -v amdgpu.ids | grep "N=   32\|K=   64" | sed -e 's/| fwd .*| dx /| dx /' | cut -c1-160
done
TSII_GEMM_PC_BNB_MIN_K=32 timeout 300 python tools/pc_check.py 2>&1 | tail -1
