#!/bin/bash
# round 3, run 39: every GPU test of tests/test_parity_ops.py at the final state
set -u
ulimit -c 0
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; mkdir -p gpurun_out
timeout 200 python -m pytest tests/test_parity_ops.py -m gpu -q > gpurun_out/r03k4c_tests3.log 2>&1; echo "rc=$?"; tail -3 gpurun_out/r03k4c_tests3.log
