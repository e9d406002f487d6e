#!/bin/bash
# round 6, call k: the row-phase kernel with rows in groups (no scratch in any form); seg + workload tests; cfg 3 line
mkdir -p gpurun_out/r06k
python tools/dw_bench.py cfg3 > gpurun_out/r06k/dw_bench_cfg3_stock.log 2>&1
grep "^dw" gpurun_out/r06k/dw_bench_cfg3_stock.log | head -4 | cut -c1-230
python -m pytest tests/test_parity_r2.py tests/test_parity_seg.py -m gpu -x -q > gpurun_out/r06k/tests_seg.log 2>&1; echo "seg tests rc=$?"; tail -2 gpurun_out/r06k/tests_seg.log
python bench.py --model TextSegament --size 512 --batch 64 --pixel-shuffle --steps 6 --warmup 2 --no-cpu-baseline --no-f32-leg 2>&1 | tail -1 > gpurun_out/r06k/bench_cfg3.json
python -c "
import json; d=json.load(open('gpurun_out/r06k/bench_cfg3.json')); print('cfg3', d['value'], d['ms_per_step'], {k:v['ms_per_step'] for k,v in d['kernel_classes'].items()})"
