#!/bin/bash
# round 3, run 1: first on-chip check + timing of the producer / consumer GEMM (gemm_pc.hip)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 300 python tools/pc_check.py > gpurun_out/r03a_pc_check.log 2>&1; echo "pc_check rc=$?"; tail -12 gpurun_out/r03a_pc_check.log
TSII_GEMM_PC=1 timeout 300 python tools/gemm_bench.py --only nt --iters 5 > gpurun_out/r03a_gemm_pc1.log 2>&1; echo "bench pc1 rc=$?"
TSII_GEMM_PC=0 timeout 300 python tools/gemm_bench.py --only nt --iters 5 > gpurun_out/r03a_gemm_pc0.log 2>&1; echo "bench pc0 rc=$?"
paste -d'\n' gpurun_out/r03a_gemm_pc1.log gpurun_out/r03a_gemm_pc0.log | grep -v amdgpu.ids
TSII_GEMM_PC=1 timeout 400 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-f32-leg > gpurun_out/r03a_bench_pc1.log 2>&1; echo "bench rc=$?"
python - <<PY
import json
for f in ("gpurun_out/r03a_bench_pc1.log",):
    try:
        l=[x for x in open(f) if x.startswith("{")][-1]
        d=json.loads(l); print(f, d["value"], d["ms_per_step"], "fwd", d.get("forward_only")); print({k:(v.get("ms"),v.get("tflops")) for k,v in d["kernel_classes"].items()})
    except Exception as e:
        print(f, "no line", e); print(open(f).read()[-2000:])
PY
