#!/bin/bash
# Collects the rocprofv3 evidence kept under profiles/ (run on the GPU box from the repo root):
#   tools/collect_profiles.sh <tag>        e.g. r01_f
# 1. bench.py line (default run, with the CPU baseline)          -> gpurun_out/<tag>_bench_bs32.log
# 2. rocprofv3 --kernel-trace --stats around bench.py --steps 3  -> gpurun_out/<tag>_kernel_stats_bs32.csv
# 3. separate counter passes (never combined with other traces): FETCH_SIZE, WRITE_SIZE, SQ_* around 2 steps
#    -> summarised by tools/pmc_summary.py into gpurun_out/<tag>_pmc_*.{csv,json}
# 4. per-entry-point / per-shape timing of one step              -> gpurun_out/<tag>_per_shape.log
# Copy the outputs into profiles/ and commit them.
set -u
ulimit -c 0      # a faulting 70 GiB process must not fill the box with its core file
TAG=${1:-r05}
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R"; mkdir -p gpurun_out
timeout 900 python bench.py > gpurun_out/${TAG}_bench_bs32.log 2>&1; tail -1 gpurun_out/${TAG}_bench_bs32.log | cut -c1-200
timeout 600 python tools/profile_step.py > gpurun_out/${TAG}_per_shape.log 2>&1
timeout 600 python tools/profile_step.py --forward > gpurun_out/${TAG}_per_shape_fwd.log 2>&1
timeout 300 python tools/dw_bench.py > gpurun_out/${TAG}_dw_bench.log 2>&1
timeout 600 python tools/host_pipeline.py --workers 1,8,16,32,64 --steps 10 2>&1 | tail -1 > gpurun_out/${TAG}_host_pipeline.log
# secondary configs (BASELINE cfg 3 / cfg 5 shapes), same bench contract, short runs
timeout 600 python bench.py --model TextSegament --batch 64 --pixel-shuffle --steps 8 --warmup 2 --no-f32-leg 2>&1 | tail -1 > gpurun_out/${TAG}_bench_cfg3_textsegament_bs64.log
timeout 600 python bench.py --model XceptionTextSegment --size 1024 --batch 8 --products 1 --steps 8 --warmup 2 --no-f32-leg 2>&1 | tail -1 > gpurun_out/${TAG}_bench_cfg5_xception1024_bf16.log
timeout 600 python bench.py --model XceptionTextSegment --size 1024 --batch 8 --steps 8 --warmup 2 --no-f32-leg 2>&1 | tail -1 > gpurun_out/${TAG}_bench_cfg5_xception1024_fp32class.log
# cfg 5 in bf16 activation storage (round 5): the line, the per-entry-point microbench, and a kernel-stats profile of the same command below
timeout 600 python bench.py --model XceptionTextSegment --size 1024 --batch 8 --storage bf16 --steps 10 --warmup 3 2>&1 | tail -1 > gpurun_out/${TAG}_bench_cfg5_xception1024_bf16storage.log
timeout 400 python tools/bf16_bench.py > gpurun_out/${TAG}_bf16_bench.log 2>&1
timeout 300 python tools/microbench.py > gpurun_out/${TAG}_microbench.log 2>&1
# evidence lines: the two Origin nets, the Bernoulli-mask stress variant, the step with the reference's full InpaintingLoss
timeout 600 python bench.py --model ImageFillOrigin --batch 16 --steps 12 --warmup 4 --no-f32-leg --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/${TAG}_bench_imagefillorigin_bs16.log
timeout 600 python bench.py --model ImageFillOriginV2 --batch 16 --steps 12 --warmup 4 --no-f32-leg --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/${TAG}_bench_imagefilloriginv2_bs16.log
timeout 600 python bench.py --bernoulli-masks --steps 10 --warmup 3 --no-f32-leg --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/${TAG}_bench_bernoulli_masks.log
timeout 600 python tools/full_loss_step.py --batch 32 --size 512 --steps 5 2>&1 | tail -1 > gpurun_out/${TAG}_full_loss_step.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/${TAG}_smoke.log 2>&1; tail -1 gpurun_out/${TAG}_smoke.log
[ "${SKIP_TESTS:-0}" = 1 ] || { timeout 1800 python -m pytest tests -m gpu -q -s > gpurun_out/${TAG}_gputests.log 2>&1; tail -1 gpurun_out/${TAG}_gputests.log; }
export TMPDIR=/tmp; cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${TAG}_prof -o b32 --output-format csv -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-f32-leg --no-secondary > $R/gpurun_out/${TAG}_rocprof.log 2>&1; echo "rocprof stats rc=$?"
cp $R/gpurun_out/${TAG}_prof/b32_kernel_stats.csv $R/gpurun_out/${TAG}_kernel_stats_bs32.csv
timeout 900 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${TAG}_prof5 -o c5 --output-format csv -- python $R/bench.py --model XceptionTextSegment --size 1024 --batch 8 --storage bf16 --steps 3 --warmup 1 --no-cpu-baseline > $R/gpurun_out/${TAG}_rocprof_cfg5.log 2>&1; echo "rocprof cfg5 rc=$?"
cp $R/gpurun_out/${TAG}_prof5/c5_kernel_stats.csv $R/gpurun_out/${TAG}_kernel_stats_cfg5_bf16storage.csv
rm -rf $R/gpurun_out/${TAG}_prof $R/gpurun_out/${TAG}_prof5   # the per-launch traces are large; the stats tables are what is kept
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --kernel-trace --pmc $c -d $R/gpurun_out/${TAG}_pmc_$c -o pmc --output-format csv -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-f32-leg --no-secondary > $R/gpurun_out/${TAG}_pmc_$c.log 2>&1; echo "pmc $c rc=$?"
done
timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $R/gpurun_out/${TAG}_pmc_sq -o pmc --output-format csv -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-f32-leg --no-secondary > $R/gpurun_out/${TAG}_pmc_sq.log 2>&1; echo "pmc sq rc=$?"
cd $R
NOTE="Each pass wraps \`python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-f32-leg --no-secondary\` (ImageFill 512x512, 32 imgs: 2 train steps + the 3-step per-class pass + 2 forward-only steps)."
python tools/pmc_summary.py hbm gpurun_out/${TAG}_pmc_FETCH_SIZE/pmc_counter_collection.csv gpurun_out/${TAG}_pmc_WRITE_SIZE/pmc_counter_collection.csv gpurun_out/${TAG}_pmc_hbm_traffic_bs32 "$NOTE"
python tools/pmc_summary.py sq gpurun_out/${TAG}_pmc_sq/pmc_counter_collection.csv gpurun_out/${TAG}_pmc_sq_bs32.csv "$NOTE"
rm -rf gpurun_out/${TAG}_pmc_FETCH_SIZE gpurun_out/${TAG}_pmc_WRITE_SIZE gpurun_out/${TAG}_pmc_sq   # raw counter dumps are large
ls -la gpurun_out | grep ${TAG}
