#!/bin/bash
# round 4, run 17: head weight gradient on the f32 matrix cores (tsii_head_cat_bwd_dw_low): parity on the chip, A/B of the step, per-shape
set -u
ulimit -c 0
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_parity_ops.py -m gpu -q -x -k "head_over_virtual or imagefill or origin" > gpurun_out/r04o_tests.log 2>&1; tail -5 gpurun_out/r04o_tests.log
for v in 0 1 0 1; do
  TSII_HEAD_MFMA=$v timeout 300 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-f32-leg 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('head_mfma=$v', d['ms_per_step'], d['value'])"
done
timeout 600 python tools/profile_step.py > gpurun_out/r04o_per_shape.log 2>&1; grep -E "head_cat|dense_bwd_dw|step total" gpurun_out/r04o_per_shape.log | head
