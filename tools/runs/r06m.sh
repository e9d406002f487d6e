#!/bin/bash
# round 6, call m: the final collection at the final sources (bench line with its secondary legs, kernel stats, PMC passes, the full GPU suite, smoke)
set -u
ulimit -c 0
TAG=r06
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R"; mkdir -p gpurun_out
timeout 900 python bench.py > gpurun_out/${TAG}_bench_bs32.log 2>&1; tail -1 gpurun_out/${TAG}_bench_bs32.log | cut -c1-160
timeout 600 python tools/profile_step.py > gpurun_out/${TAG}_per_shape.log 2>&1
timeout 600 python tools/profile_step.py --forward > gpurun_out/${TAG}_per_shape_fwd.log 2>&1
timeout 600 python bench.py --model TextSegament --batch 64 --pixel-shuffle --steps 8 --warmup 2 --no-f32-leg 2>&1 | tail -1 > gpurun_out/${TAG}_bench_cfg3_textsegament_bs64.log
timeout 600 python bench.py --model XceptionTextSegment --size 1024 --batch 8 --storage bf16 --steps 10 --warmup 3 2>&1 | tail -1 > gpurun_out/${TAG}_bench_cfg5_xception1024_bf16storage.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/${TAG}_smoke.log 2>&1; tail -1 gpurun_out/${TAG}_smoke.log
timeout 2400 python -m pytest tests -m gpu -q -s > gpurun_out/${TAG}_gputests.log 2>&1; tail -1 gpurun_out/${TAG}_gputests.log
export TMPDIR=/tmp; cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${TAG}_prof -o b32 --output-format csv -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-f32-leg --no-secondary > $R/gpurun_out/${TAG}_rocprof.log 2>&1; echo "rocprof stats rc=$?"
cp $R/gpurun_out/${TAG}_prof/b32_kernel_stats.csv $R/gpurun_out/${TAG}_kernel_stats_bs32.csv
rm -rf $R/gpurun_out/${TAG}_prof
timeout 900 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${TAG}_prof3 -o c3 --output-format csv -- python $R/bench.py --model TextSegament --batch 64 --pixel-shuffle --steps 3 --warmup 1 --no-cpu-baseline --no-f32-leg > $R/gpurun_out/${TAG}_rocprof_cfg3.log 2>&1; echo "rocprof cfg3 rc=$?"
cp $R/gpurun_out/${TAG}_prof3/c3_kernel_stats.csv $R/gpurun_out/${TAG}_kernel_stats_cfg3.csv
rm -rf $R/gpurun_out/${TAG}_prof3
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 900 rocprofv3 --kernel-trace --pmc $c -d $R/gpurun_out/${TAG}_pmc_$c -o pmc --output-format csv -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-f32-leg --no-secondary > $R/gpurun_out/${TAG}_pmc_$c.log 2>&1; echo "pmc $c rc=$?"
done
timeout 900 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $R/gpurun_out/${TAG}_pmc_sq -o pmc --output-format csv -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-f32-leg --no-secondary > $R/gpurun_out/${TAG}_pmc_sq.log 2>&1; echo "pmc sq rc=$?"
cd $R
NOTE="Each pass wraps \`python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-f32-leg --no-secondary\` (ImageFill 512x512, 32 imgs: 2 train steps + the 3-step per-class pass + 2 forward-only steps)."
python tools/pmc_summary.py hbm gpurun_out/${TAG}_pmc_FETCH_SIZE/pmc_counter_collection.csv gpurun_out/${TAG}_pmc_WRITE_SIZE/pmc_counter_collection.csv gpurun_out/${TAG}_pmc_hbm_traffic_bs32 "$NOTE"
python tools/pmc_summary.py sq gpurun_out/${TAG}_pmc_sq/pmc_counter_collection.csv gpurun_out/${TAG}_pmc_sq_bs32.csv "$NOTE"
rm -rf gpurun_out/${TAG}_pmc_FETCH_SIZE gpurun_out/${TAG}_pmc_WRITE_SIZE gpurun_out/${TAG}_pmc_sq
ls gpurun_out | grep ${TAG}_ | head -40
