#!/bin/bash
# round 4, run 5: bench line + per-shape tables (train step and forward only) with the lean strip kernel as the default
set -u
ulimit -c 0
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; mkdir -p gpurun_out
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-f32-leg > gpurun_out/r04e_bench.log 2>&1; tail -1 gpurun_out/r04e_bench.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['forward_only'], d['whole_step_roofline'])"
timeout 200 python tools/profile_step.py > gpurun_out/r04e_per_shape.log 2>&1; head -12 gpurun_out/r04e_per_shape.log
timeout 200 python tools/profile_step.py --forward > gpurun_out/r04e_per_shape_fwd.log 2>&1; head -30 gpurun_out/r04e_per_shape_fwd.log
