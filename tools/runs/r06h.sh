#!/bin/bash
# round 6, call h: the rocprofv3 passes of the collection again WITHOUT the secondary legs (call g profiled them too: the counter passes
# ran into their time limit), and the 50-step bf16-vs-fp32 storage training curves
set -u
ulimit -c 0
TAG=r06
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R"; mkdir -p gpurun_out
python -m pytest tests/test_bf16_storage.py -m gpu -x -q -s -k "trains_like" > gpurun_out/r06h_gputests_bf16_curve.log 2>&1; echo "curve test rc=$?"; grep -A 14 "50-step training curves" gpurun_out/r06h_gputests_bf16_curve.log | head -20
export TMPDIR=/tmp; cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${TAG}_prof -o b32 --output-format csv -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-f32-leg --no-secondary > $R/gpurun_out/${TAG}_rocprof.log 2>&1; echo "rocprof stats rc=$?"
cp $R/gpurun_out/${TAG}_prof/b32_kernel_stats.csv $R/gpurun_out/${TAG}_kernel_stats_bs32.csv
rm -rf $R/gpurun_out/${TAG}_prof
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 900 rocprofv3 --kernel-trace --pmc $c -d $R/gpurun_out/${TAG}_pmc_$c -o pmc --output-format csv -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-f32-leg --no-secondary > $R/gpurun_out/${TAG}_pmc_$c.log 2>&1; echo "pmc $c rc=$?"
done
timeout 900 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $R/gpurun_out/${TAG}_pmc_sq -o pmc --output-format csv -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-f32-leg --no-secondary > $R/gpurun_out/${TAG}_pmc_sq.log 2>&1; echo "pmc sq rc=$?"
cd $R
NOTE="Each pass wraps \`python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-f32-leg --no-secondary\` (ImageFill 512x512, 32 imgs: 2 train steps + the 3-step per-class pass + 2 forward-only steps)."
python tools/pmc_summary.py hbm gpurun_out/${TAG}_pmc_FETCH_SIZE/pmc_counter_collection.csv gpurun_out/${TAG}_pmc_WRITE_SIZE/pmc_counter_collection.csv gpurun_out/${TAG}_pmc_hbm_traffic_bs32 "$NOTE"
python tools/pmc_summary.py sq gpurun_out/${TAG}_pmc_sq/pmc_counter_collection.csv gpurun_out/${TAG}_pmc_sq_bs32.csv "$NOTE"
rm -rf gpurun_out/${TAG}_pmc_FETCH_SIZE gpurun_out/${TAG}_pmc_WRITE_SIZE gpurun_out/${TAG}_pmc_sq
ls -la gpurun_out | grep -E "r06_pmc|r06_kernel"
