#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; mkdir -p gpurun_out
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-f32-leg > gpurun_out/r02n_bench.log 2>&1; echo "bench rc=$?"
python - <<PY
import json
l=[x for x in open("gpurun_out/r02n_bench.log") if x.startswith("{")][-1]
d=json.loads(l); print(d["value"], d["ms_per_step"], d["forward_only"]["ms_per_step"])
for k,v in d["kernel_classes"].items(): print("   ", k, v["ms_per_step"], v["tb_per_s"])
PY
timeout 600 python tools/profile_step.py > gpurun_out/r02n_per_shape.log 2>&1; grep "tsii_dw" gpurun_out/r02n_per_shape.log | head -24
timeout 900 python -m pytest tests -m gpu -q -x -k "parity_ops or parity_seg or memory" > gpurun_out/r02n_gputests.log 2>&1; echo "gpu tests rc=$?"; tail -3 gpurun_out/r02n_gputests.log
timeout 600 python bench.py --model TextSegament --batch 64 --pixel-shuffle --steps 8 --warmup 2 --no-f32-leg 2>&1 | tail -1 > gpurun_out/r02n_bench_cfg3.log; cut -c1-400 gpurun_out/r02n_bench_cfg3.log
timeout 600 python bench.py --model XceptionTextSegment --size 1024 --batch 8 --products 1 --steps 8 --warmup 2 --no-f32-leg 2>&1 | tail -1 > gpurun_out/r02n_bench_cfg5.log; cut -c1-400 gpurun_out/r02n_bench_cfg5.log
