#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/r02e_gputests.log 2>&1; echo "gpu tests rc=$?"; grep -E "max-ratio|256 grads|memory\]|passed|failed|^FAILED|^E  " gpurun_out/r02e_gputests.log | cut -c1-260 | head -60
for k in 1 2; do
TSII_FUSE_BN_BWD=$k timeout 600 python bench.py --steps 15 --warmup 4 --no-cpu-baseline --no-f32-leg > gpurun_out/r02e_bench_bnbwd$k.log 2>&1; echo "bench bnbwd=$k rc=$?"; tail -1 gpurun_out/r02e_bench_bnbwd$k.log | python -c "
import sys, json
j = json.loads(sys.stdin.read()); print(j['value'], j['ms_per_step'], j['forward_only'], json.dumps(j['kernel_classes']))"
done
timeout 600 python tools/profile_step.py > gpurun_out/r02e_per_shape.log 2>&1; head -40 gpurun_out/r02e_per_shape.log
