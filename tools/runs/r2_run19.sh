#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; mkdir -p gpurun_out
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r02q_bench.log 2>&1; echo "bench rc=$?"
python - <<PY
import json
l=[x for x in open("gpurun_out/r02q_bench.log") if x.startswith("{")][-1]
d=json.loads(l); print(d["value"], d["ms_per_step"], d["forward_only"]["ms_per_step"], d["split3_mode"]["ms_per_step"], d["split3_mode"]["forward_ms_per_step"])
for k,v in d["kernel_classes"].items(): print("   ", k, v["ms_per_step"], v["tb_per_s"])
PY
timeout 600 python tools/profile_step.py 2>&1 | grep "tsii_dense" | head -8
timeout 600 python tools/seg_step.py --model XceptionTextSegment --batch 8 --size 1024 --steps 3 2>&1 | tail -1
timeout 600 python tools/seg_step.py --model TextSegament --batch 32 --size 512 --steps 3 2>&1 | tail -1
