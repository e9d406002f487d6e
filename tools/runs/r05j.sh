#!/bin/bash
# round 5, run j: full GPU suite after the tightened seg-256 / products=1 bars, smoke, headline + cfg 5 bench lines
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R"; mkdir -p gpurun_out/r05j
timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | grep -v amdgpu.ids | tail -15 > gpurun_out/r05j/gputests.log
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -5 > gpurun_out/r05j/smoke.log
timeout 400 python bench.py 2>/dev/null | tail -1 > gpurun_out/r05j/bench_imagefill.json
timeout 400 python bench.py --model XceptionTextSegment --size 1024 --batch 8 --storage bf16 --steps 10 --warmup 3 2>/dev/null | tail -1 > gpurun_out/r05j/bench_cfg5_bf16.json
tail -3 gpurun_out/r05j/gputests.log; cat gpurun_out/r05j/smoke.log
