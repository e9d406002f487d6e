#!/bin/bash
# round 6, session 2, run 8: K6e on the stride-2 strips (A/B against -DDX2_FOLD=0), block / workload parity on the chip
set -u; ulimit -c 0
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_parity_ops.py tests/test_parity_r2.py tests/test_workload_sizes.py -m gpu -x -q -k "pir or stride2 or one_pass or imagefill or deferred" > gpurun_out/r06aa_gputests.log 2>&1; tail -2 gpurun_out/r06aa_gputests.log
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-f32-leg --no-secondary"
for v in stock nos2fold stock nos2fold; do
  echo "variant $v" | tee -a gpurun_out/r06aa_bench_s2fold_ab.log
  if [ $v = stock ]; then timeout 600 $B 2>&1 | tail -1 | cut -c1-240 | tee -a gpurun_out/r06aa_bench_s2fold_ab.log
  else TSII_LIBRARY=$R/tools/variants/_bin/libtsii_$v.so timeout 600 $B 2>&1 | tail -1 | cut -c1-240 | tee -a gpurun_out/r06aa_bench_s2fold_ab.log; fi
done
