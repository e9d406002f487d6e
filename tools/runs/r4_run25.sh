#!/bin/bash
# round 4, run 25: small-map dilated depth-wise kernel: parity on the chip, A/B of forward / step
set -u
ulimit -c 0
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_parity_ops.py tests/test_parity_r2.py -m gpu -q -x -k "strip or dilat or golden or imagefill or rfb" > gpurun_out/r04t_tests.log 2>&1; tail -3 gpurun_out/r04t_tests.log
for v in default nosm default nosm; do
  if [ $v = default ]; then unset TSII_LIBRARY; else export TSII_LIBRARY=$R/tools/variants/_bin/libtsii_$v.so; fi
  timeout 300 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-f32-leg 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', d['ms_per_step'], d['value'], d['forward_only']['ms_per_step'], d['kernel_classes']['dw_stencil']['ms_per_step'])"
done
unset TSII_LIBRARY
timeout 600 python tools/profile_step.py --forward 2>&1 | grep -E "32, 32, 32, 1024|forward total" | cut -c1-150
