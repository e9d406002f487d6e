#!/bin/bash
# round 4, run 28: closing run at the final sources: parity tests touching the last kernel edits, smoke, bench line, kernel stats, HBM-traffic counters
set -u
ulimit -c 0
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; mkdir -p gpurun_out
TAG=r04
timeout 1500 python -m pytest tests -m gpu -q -s > gpurun_out/${TAG}_gputests.log 2>&1; tail -1 gpurun_out/${TAG}_gputests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/${TAG}_smoke.log 2>&1; tail -1 gpurun_out/${TAG}_smoke.log
timeout 900 python bench.py > gpurun_out/${TAG}_bench_bs32.log 2>&1; tail -1 gpurun_out/${TAG}_bench_bs32.log | cut -c1-200
timeout 600 python tools/profile_step.py > gpurun_out/${TAG}_per_shape.log 2>&1
timeout 600 python tools/profile_step.py --forward > gpurun_out/${TAG}_per_shape_fwd.log 2>&1
export TMPDIR=/tmp; cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${TAG}_prof -o b32 --output-format csv -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-f32-leg > $R/gpurun_out/${TAG}_rocprof.log 2>&1; echo "rocprof stats rc=$?"
cp $R/gpurun_out/${TAG}_prof/b32_kernel_stats.csv $R/gpurun_out/${TAG}_kernel_stats_bs32.csv
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --kernel-trace --pmc $c -d $R/gpurun_out/${TAG}_pmc_$c -o pmc --output-format csv -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-f32-leg > $R/gpurun_out/${TAG}_pmc_$c.log 2>&1; echo "pmc $c rc=$?"
done
cd $R
NOTE="Each pass wraps \`python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-f32-leg\` (ImageFill 512x512, 32 imgs: 2 train steps + the 3-step per-class pass + 2 forward-only steps)."
python tools/pmc_summary.py hbm gpurun_out/${TAG}_pmc_FETCH_SIZE/pmc_counter_collection.csv gpurun_out/${TAG}_pmc_WRITE_SIZE/pmc_counter_collection.csv gpurun_out/${TAG}_pmc_hbm_traffic_bs32 "$NOTE"
rm -rf gpurun_out/${TAG}_pmc_FETCH_SIZE gpurun_out/${TAG}_pmc_WRITE_SIZE gpurun_out/${TAG}_prof
