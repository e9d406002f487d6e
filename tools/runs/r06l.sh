#!/bin/bash
# round 6, call l: per-shape tables after the row-phase / scSE / pooling kernels, ranked by excess over practical floors
mkdir -p gpurun_out/r06l
python tools/profile_step.py --model TextSegament --batch 64 --pixel-shuffle --rows 400 > gpurun_out/r06l/per_shape_cfg3.log 2>&1; echo "cfg3 rc=$?"
python tools/profile_step.py --rows 400 > gpurun_out/r06l/per_shape_bs32.log 2>&1
python tools/excess.py gpurun_out/r06l/per_shape_cfg3.log 45
head -36 gpurun_out/r06l/per_shape_cfg3.log | tail -34
