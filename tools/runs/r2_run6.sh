#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_parity_r2.py tests/test_memory_savers.py -m gpu -q -k "rfb or demo or mixed or seg_nets_256 or checkpoint" > gpurun_out/r02f_tests.log 2>&1; echo "rc=$?"
grep -E "ratio|RFB|mixed bf16|memory\]|passed|failed|^FAILED|^E  " gpurun_out/r02f_tests.log | cut -c1-220 | head -60
