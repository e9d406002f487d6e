#!/bin/bash
# round 4, run 8: rocprofv3 kernel stats of the training step at the current state (where the small kernels are)
set -u
ulimit -c 0
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; mkdir -p gpurun_out
export TMPDIR=/tmp; cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r04h_prof -o b32 --output-format csv -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-f32-leg > $R/gpurun_out/r04h_rocprof.log 2>&1; echo "rocprof stats rc=$?"
cp $R/gpurun_out/r04h_prof/b32_kernel_stats.csv $R/gpurun_out/r04h_kernel_stats_bs32.csv
rm -rf $R/gpurun_out/r04h_prof
