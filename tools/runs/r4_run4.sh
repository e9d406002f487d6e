#!/bin/bash
# round 4, run 4: lean strip A/B builds -- block order, non-temporal stores / loads, more and shorter chunks
set -u
ulimit -c 0
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; mkdir -p gpurun_out
for v in default ord1 ord1nt ord1ntl ord1nt_t6k ord1nt_t12k nt_t6k; do
  echo "== $v" >> gpurun_out/r04d_dw_variants.log
  if [ $v = default ]; then unset TSII_LIBRARY; else export TSII_LIBRARY=$R/tools/variants/_bin/libtsii_$v.so; fi
  timeout 120 python tools/dw_bench.py 2>&1 | grep "^dw" >> gpurun_out/r04d_dw_variants.log
done
cat gpurun_out/r04d_dw_variants.log
