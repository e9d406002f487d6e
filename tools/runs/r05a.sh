#!/bin/bash
# GPU run r05a: first look at bf16 activation storage on the chip
set -u
ulimit -c 0
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R"; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_bf16_kernels.py -m gpu -q -x > gpurun_out/r05a_bf16_kernels.log 2>&1; tail -3 gpurun_out/r05a_bf16_kernels.log
timeout 1200 python -m pytest tests/test_bf16_storage.py -m gpu -q -s > gpurun_out/r05a_bf16_storage.log 2>&1; grep -E "bf16 storage|passed|failed|Error" gpurun_out/r05a_bf16_storage.log | tail -40
timeout 600 python bench.py --model XceptionTextSegment --size 1024 --batch 8 --storage bf16 --steps 8 --warmup 2 2>&1 | tail -1 > gpurun_out/r05a_bench_cfg5_bf16storage.log; cut -c1-400 gpurun_out/r05a_bench_cfg5_bf16storage.log
timeout 600 python bench.py --model XceptionTextSegment --size 1024 --batch 8 --products 1 --steps 8 --warmup 2 --no-f32-leg 2>&1 | tail -1 > gpurun_out/r05a_bench_cfg5_products1.log; cut -c1-300 gpurun_out/r05a_bench_cfg5_products1.log
export TMPDIR=/tmp; cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r05a_prof -o x --output-format csv -- python $R/bench.py --model XceptionTextSegment --size 1024 --batch 8 --storage bf16 --steps 3 --warmup 1 > $R/gpurun_out/r05a_rocprof.log 2>&1; echo "rocprof rc=$?"
cp $R/gpurun_out/r05a_prof/x_kernel_stats.csv $R/gpurun_out/r05a_kernel_stats_cfg5_bf16storage.csv 2>/dev/null
rm -rf $R/gpurun_out/r05a_prof
head -25 $R/gpurun_out/r05a_kernel_stats_cfg5_bf16storage.csv | cut -c1-160
