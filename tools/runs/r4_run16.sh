#!/bin/bash
# round 4, run 16: addend gradient pooled inside the BatchNorm backward (tsii_bn_act_bwd_pre_pool): tests, A/B of the step
set -u
ulimit -c 0
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_decoder_split_resolution.py tests/test_workload_sizes.py -m gpu -q -x -k "decoder or imagefill" > gpurun_out/r04n_tests.log 2>&1; tail -5 gpurun_out/r04n_tests.log
for v in 0 1 0 1; do
  TSII_FUSE_POOL_BN_BWD=$v timeout 300 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-f32-leg 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('pool_in_bn=$v', d['ms_per_step'], d['value'])"
done
