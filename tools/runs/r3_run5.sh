#!/bin/bash
# round 3, run 5: flag-synchronised producer / consumer GEMM (v3): check, ablations, layer shapes
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 300 python tools/pc_check.py > gpurun_out/r03e_pc_check.log 2>&1; echo "pc_check rc=$?"; tail -14 gpurun_out/r03e_pc_check.log
for a in 0 16 32 128 256 272 400; do
  TSII_GEMM_PC_ABL=$a timeout 120 python tools/pc_probe.py 65536 1024 1024 2>&1 | grep -v amdgpu.ids | sed -e "s/^/abl=$a /"
done
for a in 0 128 400; do
  TSII_GEMM_PC_ABL=$a timeout 120 python tools/pc_probe.py 2097152 64 256 2>&1 | grep -v amdgpu.ids | sed -e "s/^/abl=$a /"
done
TSII_GEMM_PC=0 timeout 120 python tools/pc_probe.py 65536 1024 1024 2>&1 | grep -v amdgpu.ids
TSII_GEMM_PC=1 timeout 300 python tools/gemm_bench.py --only nt --iters 5 > gpurun_out/r03e_gemm_pc1.log 2>&1; echo "bench pc1 rc=$?"
TSII_GEMM_PC=0 timeout 300 python tools/gemm_bench.py --only nt --iters 5 > gpurun_out/r03e_gemm_pc0.log 2>&1; echo "bench pc0 rc=$?"
paste -d'\n' gpurun_out/r03e_gemm_pc1.log gpurun_out/r03e_gemm_pc0.log | grep -v amdgpu.ids
