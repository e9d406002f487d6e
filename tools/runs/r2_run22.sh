#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; mkdir -p gpurun_out
timeout 900 python tests/diag/kink_probe.py > gpurun_out/r02r_kink_probe.log 2>&1; echo "rc=$?"; grep -v Warn gpurun_out/r02r_kink_probe.log | tail -12
