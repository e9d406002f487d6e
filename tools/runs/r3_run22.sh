#!/bin/bash
# round 3, run 22: wave-symmetric NT GEMM (K3w) vs producer / consumer (K3p); recipe step diagnostic
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; mkdir -p gpurun_out
TSII_GEMM_WS=1 timeout 300 python tools/pc_check.py > gpurun_out/r03r_ws_check.log 2>&1; echo "ws_check rc=$?"; tail -3 gpurun_out/r03r_ws_check.log
TSII_GEMM_WS=1 timeout 300 python tools/gemm_bench.py --only nt --iters 5 > gpurun_out/r03r_gemm_ws1.log 2>&1; echo "ws1 rc=$?"
TSII_GEMM_WS=0 timeout 300 python tools/gemm_bench.py --only nt --iters 5 > gpurun_out/r03r_gemm_ws0.log 2>&1; echo "ws0 rc=$?"
for f in ws1 ws0; do echo "== $f"; grep -v amdgpu.ids gpurun_out/r03r_gemm_$f.log | cut -c1-260; done
timeout 900 python tests/diag/seg_recipe_probe.py > gpurun_out/r03r_seg_recipe_probe.log 2>&1; echo "probe rc=$?"; grep -v amdgpu.ids gpurun_out/r03r_seg_recipe_probe.log | tail -14
