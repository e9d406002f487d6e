#!/bin/bash
# round 6, session 2, run 7: the stride-2 dX kernel's K6c input requested one step ahead (A/B against -DDX2_Y_AHEAD=0), block parity on the chip
set -u; ulimit -c 0
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_parity_ops.py tests/test_parity_r2.py tests/test_workload_sizes.py -m gpu -x -q -k "pir or stride2 or one_pass or imagefill" > gpurun_out/r06y_gputests.log 2>&1; tail -2 gpurun_out/r06y_gputests.log
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-f32-leg --no-secondary"
for v in stock noyahead stock noyahead; do
  echo "variant $v" | tee -a gpurun_out/r06y_bench_yahead_ab.log
  if [ $v = stock ]; then timeout 600 $B 2>&1 | tail -1 | cut -c1-240 | tee -a gpurun_out/r06y_bench_yahead_ab.log
  else TSII_LIBRARY=$R/tools/variants/_bin/libtsii_$v.so timeout 600 $B 2>&1 | tail -1 | cut -c1-240 | tee -a gpurun_out/r06y_bench_yahead_ab.log; fi
done
timeout 600 python tools/profile_step.py 2>&1 | grep -E "dxdw_bn |step total" | head -8 | cut -c1-170 | tee gpurun_out/r06y_per_shape_s2.log
