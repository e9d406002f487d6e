#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; mkdir -p gpurun_out
export TMPDIR=/tmp; cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r02_prof_ts -o ts --output-format csv -- python $R/bench.py --model TextSegament --batch 64 --pixel-shuffle --steps 3 --warmup 1 --no-f32-leg > $R/gpurun_out/r02_rocprof_cfg3.log 2>&1; echo "cfg3 rc=$?"
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r02_prof_xc -o xc --output-format csv -- python $R/bench.py --model XceptionTextSegment --size 1024 --batch 8 --products 1 --steps 3 --warmup 1 --no-f32-leg > $R/gpurun_out/r02_rocprof_cfg5.log 2>&1; echo "cfg5 rc=$?"
cd $R
cp gpurun_out/r02_prof_ts/ts_kernel_stats.csv gpurun_out/r02_kernel_stats_cfg3_textsegament_bs64.csv
cp gpurun_out/r02_prof_xc/xc_kernel_stats.csv gpurun_out/r02_kernel_stats_cfg5_xception1024_bf16.csv
rm -rf gpurun_out/r02_prof_ts gpurun_out/r02_prof_xc
head -12 gpurun_out/r02_kernel_stats_cfg5_xception1024_bf16.csv | cut -c1-160
