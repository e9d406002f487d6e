#!/bin/bash
# round 4, run 27b: small-map kernel ablations (no stores / no loads / one tap) on the dilation-8 layer
set -u
ulimit -c 0
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; mkdir -p gpurun_out
for v in default sma1 sma2 sma4; do
  if [ $v = default ]; then unset TSII_LIBRARY; else export TSII_LIBRARY=$R/tools/variants/_bin/libtsii_$v.so; fi
  echo "== $v"
  timeout 600 python tools/profile_step.py --forward 2>&1 | grep -E "\(32, 32, 32, 1024, 3, 3, 1, 1, 8" | cut -c1-150
done
