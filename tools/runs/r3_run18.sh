#!/bin/bash
# round 3, run 18: VALU issue-rate probe, full GPU test suite at the tightened tolerances, bench line
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; mkdir -p gpurun_out
timeout 120 tools/probes/_bin/valu_rates 20000 4 > gpurun_out/r03n_valu_rates.log 2>&1; echo "valu rc=$?"; cat gpurun_out/r03n_valu_rates.log
timeout 120 tools/probes/_bin/valu_rates 20000 1 >> gpurun_out/r03n_valu_rates.log 2>&1
timeout 1500 python -m pytest tests -m gpu -x -q -s > gpurun_out/r03n_gputests.log 2>&1; echo "gputests rc=$?"; tail -30 gpurun_out/r03n_gputests.log
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-f32-leg > gpurun_out/r03n_bench.log 2>&1; echo "bench rc=$?"
python - <<'PY'
import json
l=[x for x in open("gpurun_out/r03n_bench.log") if x.startswith("{")][-1]
d=json.loads(l); print(d["value"], d["ms_per_step"], "fwd", d.get("forward_only")); print(d["roofline"])
PY
