#!/bin/bash
# round 6, call n: TN (weight-gradient) split-M plan -- one full wave of blocks (768) against the 1024 of rounds 2-5, every dW launch of the step replayed
mkdir -p gpurun_out/r06n
for v in stock tn1024; do
  if [ $v = stock ]; then unset TSII_LIBRARY; else export TSII_LIBRARY=tools/variants/_bin/libtsii_$v.so; fi
  python tools/nt_bench.py --tn --iters 10 > gpurun_out/r06n/tn_bench_$v.log 2>&1
  echo "$v: $(tail -1 gpurun_out/r06n/tn_bench_$v.log)"
done
unset TSII_LIBRARY
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --no-f32-leg 2>&1 | tail -1 > gpurun_out/r06n/bench_bs32.json
python -c "
import json; d=json.load(open('gpurun_out/r06n/bench_bs32.json')); print('bs32', d['value'], d['ms_per_step'], {k:v['ms_per_step'] for k,v in d['kernel_classes'].items()})"
python -m pytest tests/test_parity_ops.py -m gpu -x -q 2>&1 | tail -2
