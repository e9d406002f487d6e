#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R"; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_bf16_kernels.py tests/test_bf16_storage.py -m gpu -q -s > gpurun_out/r05g_tests.log 2>&1; grep -E "bf16 storage\]|^    |passed|failed|Error|assert" gpurun_out/r05g_tests.log | grep -v "smooth:\|leaky:" | tail -50
timeout 300 python tools/bf16_bench.py --only dw,bn 2>&1 | grep -A6 "stride 2\|BatchNorm" | head -60
timeout 600 python bench.py --model XceptionTextSegment --size 1024 --batch 8 --storage bf16 --steps 8 --warmup 2 2>&1 | tail -1 > gpurun_out/r05g_bench_cfg5_bf16storage.log; cut -c1-300 gpurun_out/r05g_bench_cfg5_bf16storage.log
