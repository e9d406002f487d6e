#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; mkdir -p gpurun_out
for p in 0 1; do
  echo "== pipe=$p"
  TSII_GEMM_PIPE=$p timeout 600 python tools/gemm_bench.py --iters 5 --only fwd > gpurun_out/r02i_gemm_pipe$p.log 2>&1; echo "rc=$?"
  cat gpurun_out/r02i_gemm_pipe$p.log | tail -40
done
for p in 0 1; do
  TSII_GEMM_PIPE=$p timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-f32-leg > gpurun_out/r02i_bench_pipe$p.log 2>&1; echo "bench pipe=$p rc=$?"
  python - <<PY
import json
l=[x for x in open("gpurun_out/r02i_bench_pipe$p.log") if x.startswith("{")][-1]
d=json.loads(l); print(d["value"], d["ms_per_step"], d.get("forward_ms"))
PY
done
