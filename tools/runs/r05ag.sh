#!/bin/bash
# round 5, run ag: gathered operands of the dense convolutions on the 256 x 256 kernel (stock) vs the 128 x 256 register-staged tiles
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R"; mkdir -p gpurun_out/r05ag
timeout 600 python -m pytest tests/test_bf16_kernels.py tests/test_bf16_storage.py -m gpu -q -x 2>&1 | tail -2
for v in; do
  if [ $v = stock ]; then unset TSII_LIBRARY; else export TSII_LIBRARY=$R/tools/variants/_bin/libtsii_$v.so; fi
  echo "=== $v"; timeout 300 python tools/bf16_bench.py --only dense 2>&1 | grep -v amdgpu.ids
done > gpurun_out/r05ag/ph_conv.log 2>&1
cat gpurun_out/r05ag/ph_conv.log | grep "===\|dense\|fwd\|dX"
for v in stock nt_nophconv stock nt_nophconv; do
  if [ $v = stock ]; then unset TSII_LIBRARY; else export TSII_LIBRARY=$R/tools/variants/_bin/libtsii_$v.so; fi
  timeout 400 python bench.py --model XceptionTextSegment --size 1024 --batch 8 --storage bf16 --steps 10 --warmup 3 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('$v', d['value'], d['ms_per_step'], {k:v['ms_per_step'] for k,v in d['kernel_classes'].items()})"
done | tee -a gpurun_out/r05ag/ph_conv.log
