#!/bin/bash
# round 3, run 15: whole GPU suite (no -x) with the producer / consumer GEMM
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -q > gpurun_out/r03k_gputests.log 2>&1; echo "gpu tests rc=$?"; tail -15 gpurun_out/r03k_gputests.log
