#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_parity_r2.py tests/test_parity_ops.py -m gpu -q -k "seg_nets_256 or strip or pir or imagefill_train" > gpurun_out/r02h_tests.log 2>&1; echo "tests rc=$?"; grep -E "median|passed|failed|^FAILED|^E  " gpurun_out/r02h_tests.log | cut -c1-200 | head
for t in 0 1; do
TSII_GEMM_TILE=$t timeout 600 python bench.py --steps 15 --warmup 4 --no-cpu-baseline --no-f32-leg > gpurun_out/r02h_bench_tile$t.log 2>&1; echo "bench tile=$t rc=$?"; tail -1 gpurun_out/r02h_bench_tile$t.log | python -c "
import sys, json
j = json.loads(sys.stdin.read()); print(j['value'], j['ms_per_step'], j['forward_only']['ms_per_step']); [print('   ', c, v['ms_per_step'], v.get('tb_per_s')) for c, v in j['kernel_classes'].items()]"
done
timeout 600 python tools/profile_step.py > gpurun_out/r02h_per_shape.log 2>&1; grep -E "dw_fwd_bn" gpurun_out/r02h_per_shape.log | head -8
