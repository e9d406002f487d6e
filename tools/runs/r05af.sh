#!/bin/bash
# round 5, run ae: tile order of the 256 x 256 NT kernel inside an XCD (groups of 8 / 4 / 16 row tiles, or row-major), plain product, same box
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R"; mkdir -p gpurun_out/r05af
for v in stock nt_noprio nt_prio3 stock nt_noprio; do
  if [ $v = stock ]; then unset TSII_LIBRARY; else export TSII_LIBRARY=$R/tools/variants/_bin/libtsii_$v.so; fi
  echo "=== $v"; timeout 200 python tools/bf16_bench.py --gemm "65536,4096,4096;8192,8192,8192;131072,512,512" --iters 10 2>&1 | grep -v amdgpu.ids
done > gpurun_out/r05af/tile_order.log 2>&1
grep "===\|gemm\|launches" gpurun_out/r05af/tile_order.log | awk '/===/{v=$2} /gemm/{s=$2" "$3" "$4} /launches/{print v, s, $6, $7, $10, $11}' | sort -k2,4 -k1,1 | awk '{print}' | head -80
