#!/bin/bash
# round 4, run 13: which change moved TextSegament-256's RFB gradients (test_seg_nets_256_vs_oracle_gpu)?  A/B builds
set -u
ulimit -c 0
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; mkdir -p gpurun_out
for v in default nolean nobnsmall tn_old; do
  if [ $v = default ]; then unset TSII_LIBRARY; else export TSII_LIBRARY=$R/tools/variants/_bin/libtsii_$v.so; fi
  echo "== $v" >> gpurun_out/r04k_seg256_ab.log
  timeout 300 python -m pytest tests/test_parity_r2.py -m gpu -q -s -k "seg_nets_256_vs_oracle_gpu and TextSegament" 2>&1 | grep -E "max-ratio|passed|failed|median" | head -8 >> gpurun_out/r04k_seg256_ab.log
done
cat gpurun_out/r04k_seg256_ab.log
