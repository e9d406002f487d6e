#!/bin/bash
# round 4, run 3: strip-pattern copy probe (what the marching-strip access pattern itself delivers)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; mkdir -p gpurun_out
timeout 120 tools/probes/_bin/strip_copy > gpurun_out/r04c_strip_copy.log 2>&1; cat gpurun_out/r04c_strip_copy.log
