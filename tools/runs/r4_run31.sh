#!/bin/bash
# round 4, run 31: closing run after the fused head d low: HBM-traffic counters first (bench.py checks their source fingerprint), then
# the bench line, smoke, the parity tests of the touched paths, per-shape tables, kernel stats
set -u
ulimit -c 0
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; mkdir -p gpurun_out
TAG=r04
export TMPDIR=/tmp; cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --kernel-trace --pmc $c -d $R/gpurun_out/${TAG}_pmc_$c -o pmc --output-format csv -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-f32-leg > $R/gpurun_out/${TAG}_pmc_$c.log 2>&1; echo "pmc $c rc=$?"
done
cd $R
NOTE="Each pass wraps \`python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-f32-leg\` (ImageFill 512x512, 32 imgs: 2 train steps + the 3-step per-class pass + 2 forward-only steps)."
python tools/pmc_summary.py hbm gpurun_out/${TAG}_pmc_FETCH_SIZE/pmc_counter_collection.csv gpurun_out/${TAG}_pmc_WRITE_SIZE/pmc_counter_collection.csv gpurun_out/${TAG}_pmc_hbm_traffic_bs32 "$NOTE"
cp gpurun_out/${TAG}_pmc_hbm_traffic_bs32.json profiles/      # so that the bench line below reports the traffic of THESE sources
rm -rf gpurun_out/${TAG}_pmc_FETCH_SIZE gpurun_out/${TAG}_pmc_WRITE_SIZE
timeout 600 python bench.py > gpurun_out/${TAG}_bench_bs32.log 2>&1; tail -1 gpurun_out/${TAG}_bench_bs32.log | cut -c1-160
timeout 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/${TAG}_smoke.log 2>&1; tail -1 gpurun_out/${TAG}_smoke.log
timeout 300 python -m pytest tests/test_parity_ops.py tests/test_workload_sizes.py -m gpu -q -k "not xception and not textsegament" > gpurun_out/r04x_tests.log 2>&1; tail -1 gpurun_out/r04x_tests.log
timeout 200 python tools/profile_step.py > gpurun_out/${TAG}_per_shape.log 2>&1
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${TAG}_prof -o b32 --output-format csv -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-f32-leg > $R/gpurun_out/${TAG}_rocprof.log 2>&1; echo "rocprof stats rc=$?"
cp $R/gpurun_out/${TAG}_prof/b32_kernel_stats.csv $R/gpurun_out/${TAG}_kernel_stats_bs32.csv; rm -rf $R/gpurun_out/${TAG}_prof
