#!/bin/bash
# round 3, run 20: packed-fp32 instructions out of the MFMA kernels (A: scalar v_sub in the split only, B: no v_pk_*_f32 at all in
# gemm*.hip), producer wave priorities on top; then the full GPU suite and the bench line
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; mkdir -p gpurun_out
timeout 300 python tools/pc_check.py > gpurun_out/r03p_pc_check.log 2>&1; echo "pc_check rc=$?"; tail -2 gpurun_out/r03p_pc_check.log
TSII_LIBRARY=$R/tools/probes/_bin/libtsii_hip_scalar_sub.so timeout 300 python tools/gemm_bench.py --iters 5 > gpurun_out/r03p_gemm_A_scalar_sub.log 2>&1; echo "A rc=$?"
timeout 300 python tools/gemm_bench.py --iters 5 > gpurun_out/r03p_gemm_B_no_packed.log 2>&1; echo "B rc=$?"
for o in 4 8 12 13; do
  TSII_GEMM_PC_OPT=$o timeout 300 python tools/gemm_bench.py --only nt --iters 5 > gpurun_out/r03p_gemm_B_opt$o.log 2>&1; echo "B opt $o rc=$?"
done
for f in A_scalar_sub B_no_packed B_opt4 B_opt8 B_opt12 B_opt13; do echo "== $f"; grep -v amdgpu.ids gpurun_out/r03p_gemm_$f.log | cut -c1-260; done
timeout 1800 python -m pytest tests -m gpu -q -s > gpurun_out/r03p_gputests.log 2>&1; echo "gputests rc=$?"; grep -n "^FAILED\|^ERROR\|passed\|failed" gpurun_out/r03p_gputests.log | tail -30
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-f32-leg > gpurun_out/r03p_bench.log 2>&1; echo "bench rc=$?"
python - <<'PY'
import json
l=[x for x in open("gpurun_out/r03p_bench.log") if x.startswith("{")][-1]
d=json.loads(l); print(d["value"], d["ms_per_step"], "fwd", d.get("forward_only"))
for k,v in d["kernel_classes"].items(): print(k, v["ms_per_step"], v.get("tb_per_s"), v.get("fp32_equiv_tflops"))
PY
