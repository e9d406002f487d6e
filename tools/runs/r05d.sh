#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R"; mkdir -p gpurun_out
for v in stock abl1 abl2 abl4 abl8 abl3 abl7; do
  if [ $v = stock ]; then unset TSII_LIBRARY; else export TSII_LIBRARY=$R/tools/variants/_bin/libtsii_nt_$v.so; fi
  echo "=== $v"; timeout 120 python tools/bf16_bench.py --only pw 2>&1 | grep -A2 "K=512 N=512\|M=524288 K=128 N=128" | grep -v "^--"
done > gpurun_out/r05d_nt_ablations.log 2>&1
cat gpurun_out/r05d_nt_ablations.log
