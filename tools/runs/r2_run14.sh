#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q -x > gpurun_out/r02m_gputests.log 2>&1; echo "gpu tests rc=$?"; tail -3 gpurun_out/r02m_gputests.log
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-f32-leg > gpurun_out/r02m_bench.log 2>&1; echo "bench rc=$?"
python - <<PY
import json
l=[x for x in open("gpurun_out/r02m_bench.log") if x.startswith("{")][-1]
d=json.loads(l); print(d["value"], d["ms_per_step"], d["forward_only"]["ms_per_step"])
for k,v in d["kernel_classes"].items(): print("   ", k, v["ms_per_step"])
PY
timeout 600 python tools/seg_step.py --model TextSegament --batch 32 --size 512 --steps 3 2>&1 | tail -1 | tee gpurun_out/r02m_seg.log
timeout 600 python tools/seg_step.py --model XceptionTextSegment --batch 8 --size 1024 --steps 3 2>&1 | tail -1 | tee -a gpurun_out/r02m_seg.log
timeout 600 python tools/seg_step.py --model XceptionTextSegment --batch 8 --size 1024 --steps 3 --products 1 2>&1 | tail -1 | tee -a gpurun_out/r02m_seg.log
timeout 600 python bench.py --model ImageFillOrigin --steps 10 --warmup 3 --no-cpu-baseline --no-f32-leg 2>&1 | tail -1 | cut -c1-250 | tee -a gpurun_out/r02m_seg.log
