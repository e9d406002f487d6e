#!/bin/bash
# round 5, run k: the seg-256 fixture test alone with its table, three times (does the outlier move between runs?)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R"; mkdir -p gpurun_out/r05k
for i in 1 2; do
timeout 600 python -m pytest tests/test_parity_r2.py -m gpu -q -s -k "seg_nets_256_vs_reference_fixture or mixed_bf16_products" 2>&1 | grep -v amdgpu.ids | tail -60 > gpurun_out/r05k/seg256_$i.log
done
timeout 900 python -m pytest tests -m gpu -q --deselect "tests/test_parity_r2.py::test_seg_nets_256_vs_reference_fixture_gpu" 2>&1 | grep -v amdgpu.ids | tail -8 > gpurun_out/r05k/gputests_rest.log
tail -3 gpurun_out/r05k/gputests_rest.log
