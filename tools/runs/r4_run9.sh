#!/bin/bash
# round 4, run 9: workload-size tests (cfg 2 at bs 32, cfg 3 with its head at 512^2, cfg 5's net at 1024^2), smoke with its explanation, seg-256 fixture rule
set -u
ulimit -c 0
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; mkdir -p gpurun_out
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r04i_smoke.log 2>&1; tail -3 gpurun_out/r04i_smoke.log
timeout 1200 python -m pytest tests/test_workload_sizes.py -m gpu -q -x -s > gpurun_out/r04i_workload_tests.log 2>&1; tail -8 gpurun_out/r04i_workload_tests.log
timeout 600 python -m pytest tests/test_parity_r2.py -m gpu -q -x -s -k "seg_nets_256_vs_reference_fixture" > gpurun_out/r04i_seg256.log 2>&1; tail -25 gpurun_out/r04i_seg256.log
