#!/bin/bash
# round 3, run 19: VALU issue-rate probe (priorities, chain modes), full GPU test suite at the tightened tolerances
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; mkdir -p gpurun_out
timeout 240 tools/probes/_bin/valu_rates 2000 > gpurun_out/r03o_valu_rates.log 2>&1; echo "valu rc=$?"; cat gpurun_out/r03o_valu_rates.log
timeout 1800 python -m pytest tests -m gpu -q -s > gpurun_out/r03o_gputests.log 2>&1; echo "gputests rc=$?"; grep -n "^FAILED\|^ERROR\|passed\|failed" gpurun_out/r03o_gputests.log | tail -30
