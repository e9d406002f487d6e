#!/bin/bash
# round 3, run 9: the producers' own pace (idle consumers): 2448 = with loads, 2512 = without
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; mkdir -p gpurun_out
for a in 2448 2512 400 464; do
  TSII_GEMM_PC_ABL=$a timeout 120 python tools/pc_probe.py 65536 1024 1024 10 2>&1 | grep -v amdgpu.ids | sed -e "s/^/abl=$a /"
done
