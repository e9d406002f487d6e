#!/bin/bash
# round 4, run 30: head d low inside the weight-gradient kernel: parity on the chip, A/B of the step
set -u
ulimit -c 0
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_parity_ops.py -m gpu -q -x -k "head_over_virtual or imagefill_golden or imagefill_train" > gpurun_out/r04w_tests.log 2>&1; tail -2 gpurun_out/r04w_tests.log
for v in 0 1 0 1; do
  TSII_HEAD_DLOW=$v timeout 300 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-f32-leg 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('head_dlow=$v', d['ms_per_step'], d['value'], d['kernel_classes']['dense_conv']['ms_per_step'])"
done
