#!/bin/bash
# round 4, run 1: lean stride-1 strip kernel -- micro-benchmark, GPU parity of the strip / block / ImageFill tests, short bench line
set -u
ulimit -c 0
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; mkdir -p gpurun_out
timeout 120 python tools/dw_bench.py > gpurun_out/r04a_dw_bench.log 2>&1; cat gpurun_out/r04a_dw_bench.log | tail -5
timeout 600 python -m pytest tests/test_parity_ops.py -m gpu -q -x > gpurun_out/r04a_tests.log 2>&1; tail -3 gpurun_out/r04a_tests.log
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-f32-leg > gpurun_out/r04a_bench.log 2>&1; tail -1 gpurun_out/r04a_bench.log | cut -c1-300
timeout 200 python tools/profile_step.py > gpurun_out/r04a_per_shape.log 2>&1; head -24 gpurun_out/r04a_per_shape.log
