#!/bin/bash
# round 4, run 12: full GPU suite after restoring the sticky arithmetic-switch re-application in _lib.py
set -u
ulimit -c 0
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; mkdir -p gpurun_out
timeout 1800 python -m pytest tests -m gpu -q -s > gpurun_out/r04_gputests.log 2>&1; tail -3 gpurun_out/r04_gputests.log
