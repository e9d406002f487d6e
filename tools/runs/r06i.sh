#!/bin/bash
# round 6, call i: the row-phase depth-wise kernel (dilation >= 8 on 64-column maps) against the strip / direct kernels; seg tests; cfg 3 line
mkdir -p gpurun_out/r06i
for v in stock norows; do
  if [ $v = stock ]; then unset TSII_LIBRARY; else export TSII_LIBRARY=tools/variants/_bin/libtsii_$v.so; fi
  python tools/dw_bench.py cfg3 > gpurun_out/r06i/dw_bench_cfg3_$v.log 2>&1
  echo "== $v"; grep "^dw" gpurun_out/r06i/dw_bench_cfg3_$v.log | head -4 | cut -c1-230
done
unset TSII_LIBRARY
python -m pytest tests/test_parity_r2.py tests/test_parity_seg.py tests/test_workload_sizes.py -m gpu -x -q > gpurun_out/r06i/tests_seg.log 2>&1; echo "seg tests rc=$?"; tail -2 gpurun_out/r06i/tests_seg.log
python bench.py --model TextSegament --size 512 --batch 64 --pixel-shuffle --steps 6 --warmup 2 --no-cpu-baseline --no-f32-leg 2>&1 | tail -1 > gpurun_out/r06i/bench_cfg3.json
python -c "
import json; d=json.load(open('gpurun_out/r06i/bench_cfg3.json')); print('cfg3', d['value'], d['ms_per_step'], {k:v['ms_per_step'] for k,v in d['kernel_classes'].items()})"
