#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -q > gpurun_out/r02_gputests.log 2>&1; echo "gpu tests rc=$?"; tail -5 gpurun_out/r02_gputests.log
