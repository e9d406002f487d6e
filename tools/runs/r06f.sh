#!/bin/bash
# round 6, call f: bf16 activation storage on the lean strip kernel (H16), A/B against the marching-column kernels; tests; cfg 5 line
mkdir -p gpurun_out/r06f
for v in stock nohlean; do
  if [ $v = stock ]; then unset TSII_LIBRARY; else export TSII_LIBRARY=tools/variants/_bin/libtsii_$v.so; fi
  python tools/bf16_bench.py --only dw --iters 20 > gpurun_out/r06f/bf16_bench_dw_$v.log 2>&1
  echo "== $v"; grep -E "^dw3x3|fwd|dX" gpurun_out/r06f/bf16_bench_dw_$v.log | head -40 | cut -c1-100
done
unset TSII_LIBRARY
python -m pytest tests/test_bf16_kernels.py tests/test_bf16_storage.py -m gpu -x -q > gpurun_out/r06f/tests_bf16.log 2>&1; echo "bf16 tests rc=$?"; tail -3 gpurun_out/r06f/tests_bf16.log
python bench.py --model XceptionTextSegment --size 1024 --batch 8 --storage bf16 --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/r06f/bench_cfg5.json
python -c "
import json; d=json.load(open('gpurun_out/r06f/bench_cfg5.json')); print('cfg5', d['value'], d['ms_per_step'], d['forward_only']['ms_per_step'], {k:(v['ms_per_step'], v.get('hbm_frac')) for k,v in d['kernel_classes'].items()})"
