#!/bin/bash
# round 6, session 2, run 4: the whole GPU suite at the K6d / K6e sources, the default bench line (with the cfg 5 / cfg 3 legs), cfg 3's per-shape table
set -u; ulimit -c 0
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; mkdir -p gpurun_out
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r06s_smoke.log 2>&1; tail -1 gpurun_out/r06s_smoke.log
timeout 1800 python -m pytest tests -m gpu -q -s > gpurun_out/r06s_gputests.log 2>&1; tail -1 gpurun_out/r06s_gputests.log
timeout 900 python bench.py > gpurun_out/r06s_bench_bs32.log 2>&1; tail -1 gpurun_out/r06s_bench_bs32.log | cut -c1-200
timeout 600 python tools/profile_step.py --model TextSegament --batch 64 --pixel-shuffle > gpurun_out/r06s_per_shape_cfg3.log 2>&1; head -16 gpurun_out/r06s_per_shape_cfg3.log | cut -c1-120
