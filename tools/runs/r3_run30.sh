#!/bin/bash
# round 3, run 30: what makes the BatchNorm-fused depth-wise forward slower than the plain strips? A/B builds of dwconv.hip
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; mkdir -p gpurun_out
echo "== default"; timeout 300 python tools/dw_bench.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r03z_dw_variants.log
for v in nodeep nostats nobnin noboth w3 u1 u4; do
  echo "== $v" | tee -a gpurun_out/r03z_dw_variants.log
  TSII_LIBRARY=$R/tools/probes/_bin/libtsii_dw_$v.so timeout 300 python tools/dw_bench.py 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r03z_dw_variants.log
done
