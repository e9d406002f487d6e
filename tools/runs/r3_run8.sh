#!/bin/bash
# round 3, run 8: what about the producers' loads is slow?  912 = 400 + A loads only; 1424 = 400 + loads never waited for; 512 = full kernel, A loads only
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; mkdir -p gpurun_out
for a in 0 400 912 1424 464 512; do
  TSII_GEMM_PC_ABL=$a timeout 120 python tools/pc_probe.py 65536 1024 1024 10 2>&1 | grep -v amdgpu.ids | sed -e "s/^/abl=$a /"
done
for a in 0 400 912 1424 464 512; do
  TSII_GEMM_PC_ABL=$a timeout 120 python tools/pc_probe.py 524288 384 768 10 2>&1 | grep -v amdgpu.ids | sed -e "s/^/abl=$a /"
done
