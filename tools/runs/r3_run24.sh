#!/bin/bash
# round 3, run 24: engine clock under the GEMM load (full kernel vs the MFMA-only ablation)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; mkdir -p gpurun_out
rocm-smi --showclocks --showpower 2>&1 | head -30
for a in 0 127; do
  echo "== ws abl=$a"; TSII_GEMM_WS=1 TSII_GEMM_PC_ABL=$a timeout 120 python tools/clock_watch.py 65536 1024 1024 3 2>&1 | grep -v amdgpu.ids
done | tee gpurun_out/r03t_clock_watch.log
echo "== pc"; TSII_GEMM_WS=0 timeout 120 python tools/clock_watch.py 65536 1024 1024 3 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r03t_clock_watch.log
