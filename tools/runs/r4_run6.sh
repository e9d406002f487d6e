#!/bin/bash
# round 4, run 6: K7b (split-resolution decoder 1x1 convs) -- GPU parity, bench line, per-shape tables
set -u
ulimit -c 0
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_parity_ops.py -m gpu -q -x > gpurun_out/r04f_tests.log 2>&1; tail -3 gpurun_out/r04f_tests.log
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-f32-leg > gpurun_out/r04f_bench.log 2>&1; tail -1 gpurun_out/r04f_bench.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['forward_only'], d['whole_step_roofline'])"
TSII_FUSE_UPCAT=0 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-f32-leg > gpurun_out/r04f_bench_noup.log 2>&1; tail -1 gpurun_out/r04f_bench_noup.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('upcat materialised:', d['value'], d['ms_per_step'], d['forward_only'])"
timeout 200 python tools/profile_step.py > gpurun_out/r04f_per_shape.log 2>&1; head -30 gpurun_out/r04f_per_shape.log
timeout 200 python tools/profile_step.py --forward > gpurun_out/r04f_per_shape_fwd.log 2>&1; head -40 gpurun_out/r04f_per_shape_fwd.log
