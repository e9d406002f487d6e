#!/bin/bash
# round 3, run 37: smoke + the GPU tests that drive ImageFill through trainers / losses, after K4c
set -u
ulimit -c 0
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; mkdir -p gpurun_out
timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 | tee gpurun_out/r03_smoke.log
timeout 200 python -m pytest tests/test_training_recipe.py tests/test_parity_seg.py tests/test_abi_and_host.py -m gpu -q -k "inpainting or abi or loss" > gpurun_out/r03k4c_tests2.log 2>&1; echo "rc=$?"; tail -3 gpurun_out/r03k4c_tests2.log
