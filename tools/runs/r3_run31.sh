#!/bin/bash
# round 3, run 31: K6c epilogue with the BatchNorm input requested one band ahead
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; mkdir -p gpurun_out
timeout 300 python tools/pc_check.py 2>&1 | tail -1
timeout 300 python tools/gemm_bench.py --only nt --iters 5 > gpurun_out/r03za_gemm_dxbn_prefetch.log 2>&1
grep -v amdgpu.ids gpurun_out/r03za_gemm_dxbn_prefetch.log | sed -e 's/| fwd .*| dx /| dx /' | cut -c1-160
