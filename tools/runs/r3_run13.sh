#!/bin/bash
# round 3, run 13: trimmed producer / consumer instruction streams
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; mkdir -p gpurun_out
for a in 0 464 912 33680; do
  TSII_GEMM_PC_ABL=$a timeout 120 python tools/pc_probe.py 65536 1024 1024 10 2>&1 | grep -v amdgpu.ids | sed -e "s/^/abl=$a /"
done
for o in 4 12; do
  TSII_GEMM_PC_OPT=$o timeout 120 python tools/pc_probe.py 65536 1024 1024 10 2>&1 | grep -v amdgpu.ids | sed -e "s/^/opt=$o /"
done
timeout 300 python tools/pc_check.py > gpurun_out/r03i_pc_check.log 2>&1; echo "pc_check rc=$?"; tail -2 gpurun_out/r03i_pc_check.log
TSII_GEMM_PC=1 timeout 300 python tools/gemm_bench.py --only nt --iters 5 > gpurun_out/r03i_gemm_pc1.log 2>&1; echo "bench pc1 rc=$?"
grep -v amdgpu.ids gpurun_out/r03i_gemm_pc1.log
