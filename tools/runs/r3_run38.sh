#!/bin/bash
# round 3, run 38: the inpainting evidence lines again after K4c (Origin nets, Bernoulli masks, full InpaintingLoss)
set -u
ulimit -c 0
TAG=r03
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; mkdir -p gpurun_out
timeout 100 python bench.py --model ImageFillOrigin --batch 16 --steps 8 --warmup 2 --no-f32-leg --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/${TAG}_bench_imagefillorigin_bs16.log
timeout 100 python bench.py --model ImageFillOriginV2 --batch 16 --steps 8 --warmup 2 --no-f32-leg --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/${TAG}_bench_imagefilloriginv2_bs16.log
timeout 100 python bench.py --bernoulli-masks --steps 10 --warmup 3 --no-f32-leg --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/${TAG}_bench_bernoulli_masks.log
timeout 100 python tools/full_loss_step.py --batch 32 --size 512 --steps 5 2>&1 | tail -1 > gpurun_out/${TAG}_full_loss_step.log
for f in imagefillorigin_bs16 imagefilloriginv2_bs16 bernoulli_masks; do cut -c1-160 gpurun_out/${TAG}_bench_$f.log; done; cat gpurun_out/${TAG}_full_loss_step.log
