#!/bin/bash
# round 6, session 2, run 6: K6d on the dilated ring kernels (dilation 2 / 4 / 8) and on the small-map kernel (dilation 8): parity on the chip,
# A/B against builds without them (-DST_DWG=0 [-DSM_DWG=0]) on the headline and on cfg 3
set -u; ulimit -c 0
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_parity_ops.py tests/test_parity_r2.py tests/test_parity_seg.py -m gpu -x -q > gpurun_out/r06v_gputests_dil.log 2>&1; tail -2 gpurun_out/r06v_gputests_dil.log
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-f32-leg --no-secondary"
for v in stock nodil noring stock nodil; do
  echo "variant $v" | tee -a gpurun_out/r06v_bench_dil_ab.log
  if [ $v = stock ]; then timeout 600 $B 2>&1 | tail -1 | cut -c1-240 | tee -a gpurun_out/r06v_bench_dil_ab.log
  else TSII_LIBRARY=$R/tools/variants/_bin/libtsii_$v.so timeout 600 $B 2>&1 | tail -1 | cut -c1-240 | tee -a gpurun_out/r06v_bench_dil_ab.log; fi
done
C="python bench.py --model TextSegament --batch 64 --pixel-shuffle --steps 8 --warmup 2 --no-f32-leg --no-cpu-baseline"
for v in stock nodil stock; do
  echo "cfg3 variant $v" | tee -a gpurun_out/r06v_bench_dil_ab.log
  if [ $v = stock ]; then timeout 600 $C 2>&1 | tail -1 | cut -c1-240 | tee -a gpurun_out/r06v_bench_dil_ab.log
  else TSII_LIBRARY=$R/tools/variants/_bin/libtsii_$v.so timeout 600 $C 2>&1 | tail -1 | cut -c1-240 | tee -a gpurun_out/r06v_bench_dil_ab.log; fi
done
