#!/bin/bash
# GPU run r05b: A/B of the bf16 NT GEMM (pre-rewrite | 2 waves | 3 waves with scratch) + the other bf16 entry points
set -u
ulimit -c 0
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R"; mkdir -p gpurun_out
for v in old stock w3; do
  if [ $v = stock ]; then unset TSII_LIBRARY; else export TSII_LIBRARY=$R/tools/variants/_bin/libtsii_nt_$v.so; fi
  echo "=== NT variant $v" >> gpurun_out/r05b_bf16_bench_pw.log
  timeout 300 python tools/bf16_bench.py --only pw >> gpurun_out/r05b_bf16_bench_pw.log 2>&1
done
unset TSII_LIBRARY
timeout 300 python tools/bf16_bench.py --only dw,dense,bn > gpurun_out/r05b_bf16_bench_rest.log 2>&1
cat gpurun_out/r05b_bf16_bench_pw.log | head -150
cat gpurun_out/r05b_bf16_bench_rest.log
