#!/bin/bash
# GPU run r05c: NT v2 (global->LDS ring) microbench + correctness on the chip + cfg5 line
set -u
ulimit -c 0
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R"; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_bf16_kernels.py tests/test_bf16_storage.py -m gpu -q -x -k "not 256" > gpurun_out/r05c_tests.log 2>&1; tail -3 gpurun_out/r05c_tests.log
timeout 300 python tools/bf16_bench.py --only pw,dense,bn > gpurun_out/r05c_bf16_bench.log 2>&1; cat gpurun_out/r05c_bf16_bench.log
timeout 600 python bench.py --model XceptionTextSegment --size 1024 --batch 8 --storage bf16 --steps 8 --warmup 2 2>&1 | tail -1 > gpurun_out/r05c_bench_cfg5_bf16storage.log; cut -c1-300 gpurun_out/r05c_bench_cfg5_bf16storage.log
