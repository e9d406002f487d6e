#!/bin/bash
# round 3, run 12: do the consumers wait for the producers?  33680 = 912 + timers, 33232 = 464 + timers
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; mkdir -p gpurun_out
for a in 33680 33232; do
  TSII_GEMM_PC_ABL=$a timeout 120 python tools/pc_probe.py 65536 1024 1024 10 2>&1 | grep -v amdgpu.ids | sed -e "s/^/abl=$a /"
done
