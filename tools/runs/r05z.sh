#!/bin/bash
# round 5, run z: kernel stats of the cfg 5 bf16-storage step at the current sources
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R"; mkdir -p gpurun_out/r05z
export TMPDIR=/tmp; cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r05z/prof -o c5 --output-format csv -- python $R/bench.py --model XceptionTextSegment --size 1024 --batch 8 --storage bf16 --steps 3 --warmup 1 --no-cpu-baseline > $R/gpurun_out/r05z/run.log 2>&1
cd $R
cp $(find gpurun_out/r05z/prof -name "c5_kernel_stats.csv" | head -1) gpurun_out/r05z/kernel_stats_cfg5_bf16storage.csv
rm -rf gpurun_out/r05z/prof
tail -1 gpurun_out/r05z/run.log | cut -c1-300
