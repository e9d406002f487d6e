#!/bin/bash
# round 3, run 26: rank structure of the step-2 deviation in the segmentation recipe
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; mkdir -p gpurun_out
timeout 900 python tests/diag/seg_recipe_probe.py 2>&1 | grep -v "amdgpu.ids\|Warning\|detach\|out.append" | tee gpurun_out/r03v_seg_recipe_probe.log
