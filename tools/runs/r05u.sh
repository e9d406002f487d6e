#!/bin/bash
# round 5, run u: BatchNorm-backward apply with 8 / 4 / 1 rows per thread
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R"; mkdir -p gpurun_out/r05u
timeout 600 python -m pytest tests/test_bf16_kernels.py -m gpu -q -x -k batchnorm 2>&1 | tail -2
for v in stock bn_rpt4 bn_rpt1; do
  if [ $v = stock ]; then unset TSII_LIBRARY; else export TSII_LIBRARY=$R/tools/variants/_bin/libtsii_$v.so; fi
  echo "=== $v"; timeout 300 python tools/bf16_bench.py --only bn 2>&1 | grep -v amdgpu.ids
done > gpurun_out/r05u/bn_rpt.log 2>&1
cat gpurun_out/r05u/bn_rpt.log
unset TSII_LIBRARY
timeout 400 python bench.py --model XceptionTextSegment --size 1024 --batch 8 --storage bf16 --steps 10 --warmup 3 2>/dev/null | tail -1 > gpurun_out/r05u/bench_cfg5_bf16.json
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r05u/bench_cfg5_bf16.json').read())
print(d['value'], d['ms_per_step'], d['forward_only']['ms_per_step'], {k:v['ms_per_step'] for k,v in d['kernel_classes'].items()})
PY
