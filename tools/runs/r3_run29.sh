#!/bin/bash
# round 3, run 29: software-pipelined TN (dW) kernel vs the phase-alternating one
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; mkdir -p gpurun_out
TSII_GEMM_TN_PIPE=1 timeout 600 python -m pytest tests/test_parity_r2.py -m gpu -q -s -k "arithmetic_modes" 2>&1 | grep "gemm accuracy\] mode 6\|passed\|failed" | head
TSII_GEMM_TN_PIPE=1 timeout 300 python tools/gemm_bench.py --iters 5 > gpurun_out/r03y_gemm_tnpipe1.log 2>&1; echo "pipe1 rc=$?"
TSII_GEMM_TN_PIPE=0 timeout 300 python tools/gemm_bench.py --iters 5 > gpurun_out/r03y_gemm_tnpipe0.log 2>&1; echo "pipe0 rc=$?"
for f in tnpipe1 tnpipe0; do echo "== $f"; grep -v amdgpu.ids gpurun_out/r03y_gemm_$f.log | sed -e 's/|.*| dw/| dw/' | cut -c1-120; done
