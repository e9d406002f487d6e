#!/bin/bash
# round 3, run 16: streaming-copy recipe probe + the GPU tests that failed in run 15 (noise-based recipe bars, RFB forms)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; mkdir -p gpurun_out
tools/probes/_bin/stream_copy > gpurun_out/r03l_stream_copy.log 2>&1; cat gpurun_out/r03l_stream_copy.log
timeout 1500 python -m pytest tests -m gpu -q -s -k "recipe or rfb or checkpoint or bench_line" > gpurun_out/r03l_gputests.log 2>&1; echo "gpu tests rc=$?"; grep -E "ratio|recipe|RFB|passed|failed|^E " gpurun_out/r03l_gputests.log | head -80
