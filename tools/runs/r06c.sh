#!/bin/bash
# round 6, call c: per-shape tables of the secondary configs (cfg 3, cfg 5 in bf16 storage) and of ImageFillOrigin
mkdir -p gpurun_out/r06c
python tools/profile_step.py --model TextSegament --batch 64 --pixel-shuffle --rows 160 > gpurun_out/r06c/per_shape_cfg3.log 2>&1; echo "cfg3 rc=$?"
python tools/profile_step.py --model XceptionTextSegment --size 1024 --batch 8 --storage bf16 --rows 160 > gpurun_out/r06c/per_shape_cfg5.log 2>&1; echo "cfg5 rc=$?"
python tools/profile_step.py --model ImageFillOrigin --batch 16 --rows 60 > gpurun_out/r06c/per_shape_origin.log 2>&1; echo "origin rc=$?"
head -30 gpurun_out/r06c/per_shape_cfg3.log
