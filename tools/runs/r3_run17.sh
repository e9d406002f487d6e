#!/bin/bash
# round 3, run 17: one-vector-per-thread streaming kernels: microbench, bench line, per-shape profile
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; mkdir -p gpurun_out
timeout 300 python tools/microbench.py 2>&1 | grep -v amdgpu.ids | head -3
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-f32-leg > gpurun_out/r03m_bench.log 2>&1; echo "bench rc=$?"
python - <<'PY'
import json
l=[x for x in open("gpurun_out/r03m_bench.log") if x.startswith("{")][-1]
d=json.loads(l); print(d["value"], d["ms_per_step"], "fwd", d.get("forward_only"))
for k,v in d["kernel_classes"].items(): print(k, v["ms_per_step"], v.get("tb_per_s"), v.get("fp32_equiv_tflops"))
PY
timeout 600 python tools/profile_step.py > gpurun_out/r03m_per_shape.log 2>&1; head -24 gpurun_out/r03m_per_shape.log
