#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; mkdir -p gpurun_out
for p in 0 2 3; do
  echo "== persist=$p"
  for o in fwdbn fwd dx; do
    TSII_GEMM_PERSIST=$p timeout 600 python tools/gemm_bench.py --iters 5 --only $o > gpurun_out/r02j_gemm_${o}_persist$p.log 2>&1; echo "rc=$?"
    grep mode gpurun_out/r02j_gemm_${o}_persist$p.log | sed -E 's/masked=. alg +[0-9.]+ GB//'
  done
done
for p in 0 2 3; do
  TSII_GEMM_PERSIST=$p timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-f32-leg > gpurun_out/r02j_bench_persist$p.log 2>&1; echo "bench persist=$p rc=$?"
  python - <<PY
import json
l=[x for x in open("gpurun_out/r02j_bench_persist$p.log") if x.startswith("{")][-1]
d=json.loads(l); print(d["value"], d["ms_per_step"], d["forward_only"]["ms_per_step"])
for k,v in d["kernel_classes"].items(): print("   ", k, v["ms_per_step"])
PY
done
