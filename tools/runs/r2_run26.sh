#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; mkdir -p gpurun_out
timeout 600 python bench.py --model XceptionTextSegment --size 1024 --batch 8 --products 1 --steps 8 --warmup 2 --no-f32-leg 2>&1 | tail -1 > gpurun_out/r02_bench_cfg5_xception1024_bf16.log; cut -c1-330 gpurun_out/r02_bench_cfg5_xception1024_bf16.log
timeout 600 python bench.py --model XceptionTextSegment --size 1024 --batch 8 --steps 8 --warmup 2 --no-f32-leg 2>&1 | tail -1 > gpurun_out/r02_bench_cfg5_xception1024_fp32class.log; cut -c1-330 gpurun_out/r02_bench_cfg5_xception1024_fp32class.log
timeout 600 python bench.py --model TextSegament --batch 64 --pixel-shuffle --steps 8 --warmup 2 --no-f32-leg 2>&1 | tail -1 > gpurun_out/r02_bench_cfg3_textsegament_bs64.log; cut -c1-330 gpurun_out/r02_bench_cfg3_textsegament_bs64.log
timeout 2400 python -m pytest tests -m gpu -q > gpurun_out/r02_gputests.log 2>&1; echo "gpu tests rc=$?"; tail -5 gpurun_out/r02_gputests.log
