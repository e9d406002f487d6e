#!/bin/bash
# round 3, run 42: the rest of the GPU suite at the final state, as far as the remaining GPU minutes go
set -u
ulimit -c 0
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; mkdir -p gpurun_out
timeout 160 python -m pytest tests/test_parity_r2.py tests/test_training_recipe.py -m gpu -q -k "not seg_nets_256_vs_oracle" > gpurun_out/r03k4c_tests5.log 2>&1; echo "rc=$?"; tail -4 gpurun_out/r03k4c_tests5.log
