#!/bin/bash
# round 3, run 10: load SHAPE probe: 5008 = 912 (MFMA-only consumers, A loads only) with full-line loads; 4096 = full kernel with full-line loads
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; mkdir -p gpurun_out
for a in 912 5008 0 4096 464; do
  TSII_GEMM_PC_ABL=$a timeout 120 python tools/pc_probe.py 65536 1024 1024 10 2>&1 | grep -v amdgpu.ids | sed -e "s/^/abl=$a /"
done
for a in 912 5008 0 4096 464; do
  TSII_GEMM_PC_ABL=$a timeout 120 python tools/pc_probe.py 524288 384 768 10 2>&1 | grep -v amdgpu.ids | sed -e "s/^/abl=$a /"
done
