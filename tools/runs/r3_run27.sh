#!/bin/bash
# round 3, run 27: full GPU suite at the current state; evidence lines (ImageFillOrigin / V2 steps, Bernoulli masks, full InpaintingLoss step)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; mkdir -p gpurun_out
timeout 1800 python -m pytest tests -m gpu -q -s > gpurun_out/r03w_gputests.log 2>&1; echo "gputests rc=$?"; grep -n "^FAILED\|^ERROR\|passed\|failed" gpurun_out/r03w_gputests.log | tail -20
timeout 600 python bench.py --model ImageFillOrigin --batch 16 --steps 8 --warmup 2 --no-f32-leg --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/r03_bench_imagefillorigin_bs16.log; cut -c1-300 gpurun_out/r03_bench_imagefillorigin_bs16.log
timeout 600 python bench.py --model ImageFillOriginV2 --batch 16 --steps 8 --warmup 2 --no-f32-leg --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/r03_bench_imagefilloriginv2_bs16.log; cut -c1-300 gpurun_out/r03_bench_imagefilloriginv2_bs16.log
timeout 600 python bench.py --bernoulli-masks --steps 10 --warmup 3 --no-f32-leg --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/r03_bench_bernoulli_masks.log; cut -c1-300 gpurun_out/r03_bench_bernoulli_masks.log
timeout 600 python tools/full_loss_step.py --batch 32 --size 512 --steps 5 > gpurun_out/r03_full_loss_step.log 2>&1; tail -5 gpurun_out/r03_full_loss_step.log
