#!/bin/bash
# round 3, run 14: GPU test suite + bench line with the producer / consumer GEMM
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r03j_gputests.log 2>&1; echo "gpu tests rc=$?"; tail -5 gpurun_out/r03j_gputests.log
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-f32-leg > gpurun_out/r03j_bench.log 2>&1; echo "bench rc=$?"
python - <<'PY'
import json
l=[x for x in open("gpurun_out/r03j_bench.log") if x.startswith("{")][-1]
d=json.loads(l); print(d["value"], d["ms_per_step"], "fwd", d.get("forward_only")); print(d["roofline"])
for k,v in d["kernel_classes"].items(): print(k, v)
PY
timeout 600 python tools/profile_step.py > gpurun_out/r03j_per_shape.log 2>&1; head -32 gpurun_out/r03j_per_shape.log
