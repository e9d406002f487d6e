#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; mkdir -p gpurun_out
timeout 600 python __graft_entry__.py smoke > gpurun_out/r02_smoke.log 2>&1; echo "smoke rc=$?"; tail -2 gpurun_out/r02_smoke.log
bash tools/collect_profiles.sh r02 2>&1 | tail -25
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/r02_gputests.log 2>&1; echo "gpu tests rc=$?"; tail -3 gpurun_out/r02_gputests.log
