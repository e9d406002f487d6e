#!/bin/bash
# round 6, session 2, run 2: K6d on the stride-2 dX strips too -- parity on the chip, A/B of the headline step, per-shape table
set -u; ulimit -c 0
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_parity_ops.py tests/test_parity_r2.py -m gpu -x -q > gpurun_out/r06q_gputests_k6d.log 2>&1; tail -3 gpurun_out/r06q_gputests_k6d.log
for k in 1 0 1; do
  timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-f32-leg --no-secondary --knob FUSE_DW_DXDW=$k 2>&1 | tail -1 | cut -c1-240 | tee -a gpurun_out/r06q_bench_k6d_ab.log
done
timeout 600 python tools/profile_step.py > gpurun_out/r06q_per_shape.log 2>&1; head -40 gpurun_out/r06q_per_shape.log | cut -c1-160
