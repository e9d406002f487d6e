#!/bin/bash
# round 3, run 36: final evidence after K4c (short form of tools/collect_profiles.sh: bench line, per-shape, kernel stats, HBM counters)
set -u
ulimit -c 0
TAG=r03
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; mkdir -p gpurun_out
timeout 400 python bench.py > gpurun_out/${TAG}_bench_bs32.log 2>&1; tail -1 gpurun_out/${TAG}_bench_bs32.log | cut -c1-200
timeout 200 python tools/profile_step.py > gpurun_out/${TAG}_per_shape.log 2>&1
export TMPDIR=/tmp; cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${TAG}_prof -o b32 --output-format csv -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-f32-leg > $R/gpurun_out/${TAG}_rocprof.log 2>&1; echo "rocprof stats rc=$?"
cp $R/gpurun_out/${TAG}_prof/b32_kernel_stats.csv $R/gpurun_out/${TAG}_kernel_stats_bs32.csv
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --kernel-trace --pmc $c -d $R/gpurun_out/${TAG}_pmc_$c -o pmc --output-format csv -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-f32-leg > $R/gpurun_out/${TAG}_pmc_$c.log 2>&1; echo "pmc $c rc=$?"
done
cd $R
NOTE="Each pass wraps \`python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-f32-leg\` (ImageFill 512x512, 32 imgs: 2 train steps + the 3-step per-class pass + 2 forward-only steps)."
python tools/pmc_summary.py hbm gpurun_out/${TAG}_pmc_FETCH_SIZE/pmc_counter_collection.csv gpurun_out/${TAG}_pmc_WRITE_SIZE/pmc_counter_collection.csv gpurun_out/${TAG}_pmc_hbm_traffic_bs32 "$NOTE"
rm -rf gpurun_out/${TAG}_pmc_FETCH_SIZE gpurun_out/${TAG}_pmc_WRITE_SIZE
ls -la gpurun_out | grep ${TAG}_ | head -20
