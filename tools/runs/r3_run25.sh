#!/bin/bash
# round 3, run 25: does state leak from one training step into the next?
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; mkdir -p gpurun_out
timeout 600 python tests/diag/repeat_probe.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r03u_repeat_probe.log
