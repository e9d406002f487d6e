#!/bin/bash
# round 3, run 23: ablations of the wave-symmetric kernel on one MFMA-bound shape; epilogue with prefetched row factors
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; mkdir -p gpurun_out
for a in 0 1 2 3 4 8 16 32 64 128 7 15 31 63 127 71 79 95 135; do
  TSII_GEMM_WS=1 TSII_GEMM_PC_ABL=$a timeout 120 python tools/pc_probe.py 65536 1024 1024 10 2>&1 | grep -v amdgpu.ids | sed -e "s/^/ws abl=$a /"
done | tee gpurun_out/r03s_ws_ablations.log
TSII_GEMM_WS=1 timeout 300 python tools/pc_check.py 2>&1 | tail -1
TSII_GEMM_WS=1 timeout 300 python tools/gemm_bench.py --only nt --iters 5 > gpurun_out/r03s_gemm_ws1.log 2>&1; echo "ws1 rc=$?"
TSII_GEMM_WS=0 timeout 300 python tools/gemm_bench.py --only nt --iters 5 > gpurun_out/r03s_gemm_ws0.log 2>&1; echo "ws0 rc=$?"
for f in ws1 ws0; do echo "== $f"; grep -v amdgpu.ids gpurun_out/r03s_gemm_$f.log | cut -c1-260; done
