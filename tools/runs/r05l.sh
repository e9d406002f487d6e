#!/bin/bash
# round 5, run l: the seg-256 fixture test's table (XceptionTextSegment), untruncated
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R"; mkdir -p gpurun_out/r05l
timeout 600 python -m pytest tests/test_parity_r2.py -m gpu -q -s -k "seg_nets_256_vs_reference_fixture or seg_nets_256_vs_oracle" 2>&1 | grep -v amdgpu.ids | grep "ratio\|outlier\|\[\|passed\|failed\|median" | cut -c1-250 > gpurun_out/r05l/seg256.log
cat gpurun_out/r05l/seg256.log | tail -40
