#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R"; mkdir -p gpurun_out
for v in stock dw_w2; do
  if [ $v = stock ]; then unset TSII_LIBRARY; else export TSII_LIBRARY=$R/tools/variants/_bin/libtsii_$v.so; fi
  echo "=== $v"; timeout 200 python tools/dw_bench.py 2>&1 | grep -v amdgpu.ids
  timeout 400 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-f32-leg 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('ImageFill', d['value'], d['ms_per_step'], d['forward_only']['ms_per_step'], d['kernel_classes']['dw_stencil']['ms_per_step'])"
done
for v in stock head_w2; do
  if [ $v = stock ]; then unset TSII_LIBRARY; else export TSII_LIBRARY=$R/tools/variants/_bin/libtsii_$v.so; fi
  echo "=== $v"
  timeout 400 python bench.py --model ImageFillOrigin --batch 16 --steps 12 --warmup 4 --no-f32-leg --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('Origin', d['value'], d['ms_per_step'], d['kernel_classes']['dense_conv']['ms_per_step'])"
  timeout 400 python bench.py --bernoulli-masks --steps 10 --warmup 3 --no-f32-leg --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('Bernoulli', d['value'], d['ms_per_step'], d['kernel_classes']['dense_conv']['ms_per_step'])"
done
