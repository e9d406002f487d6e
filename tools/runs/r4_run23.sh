#!/bin/bash
# round 4, run 23: ImageFillOriginV2 bs 16 with / without the matrix-core head (c1 = 64 instantiations)
set -u
ulimit -c 0
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; mkdir -p gpurun_out
for v in 1 0 1 0; do
  TSII_HEAD_MFMA=$v timeout 300 python bench.py --model ImageFillOriginV2 --batch 16 --steps 12 --warmup 4 --no-f32-leg --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('head_mfma=$v', d['ms_per_step'], d['value'], d['forward_only']['ms_per_step'], d['kernel_classes']['dense_conv']['ms_per_step'])"
done
