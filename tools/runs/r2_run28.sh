#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; mkdir -p gpurun_out
timeout 600 python bench.py --graph --steps 30 --warmup 5 --no-cpu-baseline --no-f32-leg > gpurun_out/r02s_bench_graph.log 2>&1; echo "rc=$?"
python - <<PY
import json
l=[x for x in open("gpurun_out/r02s_bench_graph.log") if x.startswith("{")][-1]
d=json.loads(l); print(d["launch"], d["value"], d["ms_per_step"], "eager", d["eager_ms_per_step"])
PY
grep -i "graph\|error" gpurun_out/r02s_bench_graph.log | head -5
