#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; mkdir -p gpurun_out
export TMPDIR=/tmp; cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r02k_prof_ts -o ts --output-format csv -- python $R/tools/seg_step.py --model TextSegament --batch 32 --size 512 --steps 3 > $R/gpurun_out/r02k_ts.log 2>&1; echo "ts rc=$?"; tail -1 $R/gpurun_out/r02k_ts.log
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r02k_prof_xc -o xc --output-format csv -- python $R/tools/seg_step.py --model XceptionTextSegment --batch 8 --size 1024 --steps 3 > $R/gpurun_out/r02k_xc.log 2>&1; echo "xc rc=$?"; tail -1 $R/gpurun_out/r02k_xc.log
cd $R
cp gpurun_out/r02k_prof_ts/ts_kernel_stats.csv gpurun_out/r02k_textsegament_kernel_stats_bs32.csv
cp gpurun_out/r02k_prof_xc/xc_kernel_stats.csv gpurun_out/r02k_xception1024_kernel_stats_bs8.csv
rm -rf gpurun_out/r02k_prof_ts gpurun_out/r02k_prof_xc
head -25 gpurun_out/r02k_textsegament_kernel_stats_bs32.csv | cut -c1-200
