#!/bin/bash
# round 6, session 2, run 3: K6e (BatchNorm backward applied on load in the one-pass depth-wise dX + dW kernel) -- parity on the chip,
# A/B of the headline step, per-shape table; the two register-budget variants of run 2's kernels
set -u; ulimit -c 0
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_parity_ops.py tests/test_parity_r2.py -m gpu -x -q > gpurun_out/r06r_gputests_k6e.log 2>&1; tail -3 gpurun_out/r06r_gputests_k6e.log
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-f32-leg --no-secondary"
for k in 1 0 1 0; do
  timeout 600 $B --knob FUSE_DW_BN2_FOLD=$k 2>&1 | tail -1 | cut -c1-240 | tee -a gpurun_out/r06r_bench_k6e_ab.log
done
for v in dx2w2 lsw2; do
  echo "variant $v" | tee -a gpurun_out/r06r_bench_k6e_ab.log
  TSII_LIBRARY=$R/tools/variants/_bin/libtsii_$v.so timeout 600 $B --knob FUSE_DW_BN2_FOLD=0 2>&1 | tail -1 | cut -c1-240 | tee -a gpurun_out/r06r_bench_k6e_ab.log
done
timeout 600 python tools/profile_step.py > gpurun_out/r06r_per_shape.log 2>&1; head -44 gpurun_out/r06r_per_shape.log | cut -c1-160
