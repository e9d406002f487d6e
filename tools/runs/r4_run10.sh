#!/bin/bash
# round 4, run 10: TN 128x64 tiles for tall x 64 dW products (A/B), host pipeline numbers, secondary configs at the current state
set -u
ulimit -c 0
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; mkdir -p gpurun_out
for v in default tn_old; do
  if [ $v = default ]; then unset TSII_LIBRARY; else export TSII_LIBRARY=$R/tools/variants/_bin/libtsii_$v.so; fi
  echo "== $v" >> gpurun_out/r04j_tn.log
  timeout 200 python tools/profile_step.py 2>&1 | grep -E "step total|tsii_pw_bwd_dw " | head -12 >> gpurun_out/r04j_tn.log
done
unset TSII_LIBRARY
cat gpurun_out/r04j_tn.log
timeout 400 python tools/host_pipeline.py --workers 1,8,16,32,64 --steps 10 > gpurun_out/r04j_host_pipeline.log 2>&1; tail -1 gpurun_out/r04j_host_pipeline.log
timeout 300 python bench.py --model TextSegament --batch 64 --pixel-shuffle --steps 6 --warmup 2 --no-f32-leg 2>&1 | tail -1 > gpurun_out/r04j_bench_cfg3.log; python -c "import json; d=json.load(open('gpurun_out/r04j_bench_cfg3.log')); print('cfg3', d['value'], d['ms_per_step'])"
timeout 300 python bench.py --model XceptionTextSegment --size 1024 --batch 8 --products 1 --steps 6 --warmup 2 --no-f32-leg 2>&1 | tail -1 > gpurun_out/r04j_bench_cfg5_bf16.log; python -c "import json; d=json.load(open('gpurun_out/r04j_bench_cfg5_bf16.log')); print('cfg5 bf16 operands', d['value'], d['ms_per_step'])"
timeout 300 python bench.py --model ImageFillOrigin --batch 16 --steps 6 --warmup 2 --no-f32-leg --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/r04j_bench_origin.log; python -c "import json; d=json.load(open('gpurun_out/r04j_bench_origin.log')); print('origin', d['value'], d['ms_per_step'])"
