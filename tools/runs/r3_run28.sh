#!/bin/bash
# round 3, run 28: strip kernels with LDS-only barriers: per-shape depth-wise bench, bench line, the depth-wise / block parity tests
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; mkdir -p gpurun_out
timeout 300 python tools/dw_bench.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r03x_dw_bench.log
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-f32-leg > gpurun_out/r03x_bench.log 2>&1; echo "bench rc=$?"
python - <<'PY'
import json
l=[x for x in open("gpurun_out/r03x_bench.log") if x.startswith("{")][-1]
d=json.loads(l); print(d["value"], d["ms_per_step"], "fwd", d.get("forward_only"))
for k,v in d["kernel_classes"].items(): print(k, v["ms_per_step"], v.get("tb_per_s"), v.get("fp32_equiv_tflops"), v.get("per_launch_roofline_frac"))
PY
timeout 900 python -m pytest tests/test_parity_ops.py tests/test_parity_r2.py -m gpu -q -x > gpurun_out/r03x_gputests.log 2>&1; echo "gputests rc=$?"; tail -3 gpurun_out/r03x_gputests.log
