#!/bin/bash
# round 3, run 40: segmentation parity + memory-saver GPU tests at the final state
set -u
ulimit -c 0
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; mkdir -p gpurun_out
timeout 150 python -m pytest tests/test_parity_seg.py tests/test_memory_savers.py -m gpu -q > gpurun_out/r03k4c_tests4.log 2>&1; echo "rc=$?"; tail -3 gpurun_out/r03k4c_tests4.log
