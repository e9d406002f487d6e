#!/bin/bash
# round 5, run ab: cfg 5 step, depth-wise kernels with the per-form two-column dispatch (stock) vs one column everywhere (dw_nx1), alternating in one call
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R"; mkdir -p gpurun_out/r05ab
for v in stock dw_nx1 stock dw_nx1; do
  if [ $v = stock ]; then unset TSII_LIBRARY; else export TSII_LIBRARY=$R/tools/variants/_bin/libtsii_$v.so; fi
  timeout 400 python bench.py --model XceptionTextSegment --size 1024 --batch 8 --storage bf16 --steps 12 --warmup 3 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('$v', d['value'], d['ms_per_step'], d['forward_only']['ms_per_step'], {k:v['ms_per_step'] for k,v in d['kernel_classes'].items()})"
done | tee gpurun_out/r05ab/dw_nx_step.log
