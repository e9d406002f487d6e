#!/bin/bash
# round 4, run 24: lean stride-2 forward strip: A/B on the depth-wise bench shapes and on the step, parity tests that cover it
set -u
ulimit -c 0
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; mkdir -p gpurun_out
for v in default nol2; do
  if [ $v = default ]; then unset TSII_LIBRARY; else export TSII_LIBRARY=$R/tools/variants/_bin/libtsii_$v.so; fi
  echo "== $v"; timeout 300 python tools/dw_bench.py 2>&1 | grep "s2"
done
unset TSII_LIBRARY
timeout 900 python -m pytest tests/test_parity_ops.py tests/test_parity_r2.py -m gpu -q -x -k "strip or stride2 or golden or imagefill" > gpurun_out/r04s_tests.log 2>&1; tail -3 gpurun_out/r04s_tests.log
for v in default nol2 default nol2; do
  if [ $v = default ]; then unset TSII_LIBRARY; else export TSII_LIBRARY=$R/tools/variants/_bin/libtsii_$v.so; fi
  timeout 300 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-f32-leg 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', d['ms_per_step'], d['value'], d['forward_only']['ms_per_step'])"
done
