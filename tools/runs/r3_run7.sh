#!/bin/bash
# round 3, run 7: scalar-base producer addressing: ablations + layer shapes
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; mkdir -p gpurun_out
export TMPDIR=/tmp
for a in 0 400 432 464 96 128; do
  TSII_GEMM_PC_ABL=$a timeout 120 python tools/pc_probe.py 65536 1024 1024 10 2>&1 | grep -v amdgpu.ids | sed -e "s/^/abl=$a /"
done
TSII_GEMM_PC=0 timeout 120 python tools/pc_probe.py 65536 1024 1024 10 2>&1 | grep -v amdgpu.ids
timeout 300 python tools/pc_check.py > gpurun_out/r03g_pc_check.log 2>&1; echo "pc_check rc=$?"; tail -2 gpurun_out/r03g_pc_check.log
TSII_GEMM_PC=1 timeout 300 python tools/gemm_bench.py --only nt --iters 5 > gpurun_out/r03g_gemm_pc1.log 2>&1; echo "bench pc1 rc=$?"
grep -v amdgpu.ids gpurun_out/r03g_gemm_pc1.log
