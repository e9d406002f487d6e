#!/bin/bash
# round 4, run 7: bn_finalize fast path + K7b addend prefetch -- bench line and forward per-shape table
set -u
ulimit -c 0
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; mkdir -p gpurun_out
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-f32-leg > gpurun_out/r04g_bench.log 2>&1; tail -1 gpurun_out/r04g_bench.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['forward_only'], d['whole_step_roofline'])"
timeout 200 python tools/profile_step.py --forward > gpurun_out/r04g_per_shape_fwd.log 2>&1; head -32 gpurun_out/r04g_per_shape_fwd.log; grep bn_finalize gpurun_out/r04g_per_shape_fwd.log | head -8
timeout 600 python -m pytest tests/test_parity_ops.py -m gpu -q -x -k "pir or imagefill_golden or imagefill_train" > gpurun_out/r04g_tests.log 2>&1; tail -3 gpurun_out/r04g_tests.log
