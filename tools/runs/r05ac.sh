#!/bin/bash
# round 5, run ac: fp32 BatchNorm-backward apply with 4 / 2 / 1 rows per thread, headline step, alternating in one call
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R"; mkdir -p gpurun_out/r05ac

for v in stock bn_rpt1_f32 stock bn_rpt1_f32; do
  if [ $v = stock ]; then unset TSII_LIBRARY; else export TSII_LIBRARY=$R/tools/variants/_bin/libtsii_$v.so; fi
  timeout 400 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-f32-leg 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('$v', d['value'], d['ms_per_step'], d['forward_only']['ms_per_step'], {k:v['ms_per_step'] for k,v in d['kernel_classes'].items()})"
done | tee gpurun_out/r05ac/bn_apply_rpt_f32.log
