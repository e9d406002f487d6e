#!/bin/bash
# round 5, run x: K6c epilogue vs plain 256 x 256 dX + stand-alone reduction on the large 1x1 layers (cfg 5 step, one call)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R"; mkdir -p gpurun_out/r05x
for k in 1 0 1 0; do
timeout 400 python bench.py --model XceptionTextSegment --size 1024 --batch 8 --storage bf16 --steps 10 --warmup 3 --knob BF16_UNFUSE_K6C_ON_LARGE=$k 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('unfuse=$k', d['value'], d['ms_per_step'], {k:v['ms_per_step'] for k,v in d['kernel_classes'].items()})"
done | tee gpurun_out/r05x/k6c_unfuse.log
timeout 600 python -m pytest tests/test_bf16_storage.py -m gpu -q -x 2>&1 | tail -2
