#!/bin/bash
# round 5, run v: BatchNorm-on-load constants from LDS in the 128 x 256 NT form
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R"; mkdir -p gpurun_out/r05v
timeout 600 python -m pytest tests/test_bf16_kernels.py tests/test_bf16_storage.py -m gpu -q -x 2>&1 | tail -2
timeout 300 python tools/bf16_bench.py --only pw 2>&1 | grep "^1x1\|fwd" > gpurun_out/r05v/pw_fwd.log; cat gpurun_out/r05v/pw_fwd.log
timeout 400 python bench.py --model XceptionTextSegment --size 1024 --batch 8 --storage bf16 --steps 10 --warmup 3 2>/dev/null | tail -1 > gpurun_out/r05v/bench_cfg5_bf16.json
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r05v/bench_cfg5_bf16.json').read())
print(d['value'], d['ms_per_step'], d['forward_only']['ms_per_step'], {k:v['ms_per_step'] for k,v in d['kernel_classes'].items()})
PY
