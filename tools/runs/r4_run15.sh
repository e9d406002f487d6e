#!/bin/bash
# round 4, run 15: seg-256 oracle test with the per-module noise floor
set -u
ulimit -c 0
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_parity_r2.py -m gpu -q -s -k "seg_nets_256_vs_oracle_gpu" > gpurun_out/r04m_seg256.log 2>&1; grep -E "max-ratio|passed|failed|median|Error" gpurun_out/r04m_seg256.log | head -30
