#!/bin/bash
# round 3, run 35: K4c (head over the virtual concatenation): parity on the chip, ImageFill tests, bench line
set -u
ulimit -c 0
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_parity_ops.py -m gpu -q -k "virtual_concat or imagefill" > gpurun_out/r03k4c_tests.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/r03k4c_tests.log
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-f32-leg > gpurun_out/r03k4c_bench.log 2>&1; echo "bench rc=$?"
python - <<'PY'
import json
l=[x for x in open("gpurun_out/r03k4c_bench.log") if x.startswith("{")][-1]
d=json.loads(l); print(d["value"], d["ms_per_step"], "fwd", d.get("forward_only"))
for k,v in d["kernel_classes"].items(): print(k, v["ms_per_step"], v.get("tb_per_s"), v.get("per_launch_roofline_frac"))
PY
timeout 300 python tools/profile_step.py 2>&1 | grep "head_cat\|dense\|upcat" | head -12
