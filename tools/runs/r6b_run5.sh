#!/bin/bash
# round 6, session 2, run 5: 32 x 128 tiles for the <= 32-row weight-gradient GEMMs (A/B against a build without them), GEMM parity on the chip
set -u; ulimit -c 0
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_parity_ops.py tests/test_parity_r2.py tests/test_parity_seg.py -m gpu -x -q > gpurun_out/r06t_gputests_thin.log 2>&1; tail -2 gpurun_out/r06t_gputests_thin.log
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-f32-leg --no-secondary"
for v in stock nothin stock nothin; do
  echo "variant $v" | tee -a gpurun_out/r06t_bench_thin_ab.log
  if [ $v = stock ]; then timeout 600 $B 2>&1 | tail -1 | cut -c1-240 | tee -a gpurun_out/r06t_bench_thin_ab.log
  else TSII_LIBRARY=$R/tools/variants/_bin/libtsii_$v.so timeout 600 $B 2>&1 | tail -1 | cut -c1-240 | tee -a gpurun_out/r06t_bench_thin_ab.log; fi
done
timeout 600 python tools/profile_step.py 2>&1 | grep -E "pw_bwd_dw|step total" | head -12 | cut -c1-160 | tee gpurun_out/r06t_per_shape_tn.log
TSII_LIBRARY=$R/tools/variants/_bin/libtsii_nothin.so timeout 600 python tools/profile_step.py 2>&1 | grep -E "pw_bwd_dw|step total" | head -12 | cut -c1-160 | tee -a gpurun_out/r06t_per_shape_tn.log
