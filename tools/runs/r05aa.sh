#!/bin/bash
# round 5, run aa: depth-wise kernels with two output columns per thread at stride 1 (stock) vs one (dw_nx1)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R"; mkdir -p gpurun_out/r05aa
timeout 600 python -m pytest tests/test_bf16_kernels.py tests/test_bf16_storage.py -m gpu -q -x 2>&1 | tail -2
for v in stock dw_nx1; do
  if [ $v = stock ]; then unset TSII_LIBRARY; else export TSII_LIBRARY=$R/tools/variants/_bin/libtsii_$v.so; fi
  echo "=== $v"; timeout 300 python tools/bf16_bench.py --only dw 2>&1 | grep -v amdgpu.ids
done > gpurun_out/r05aa/dw_nx.log 2>&1
python - <<'PY'
import re
cur=None; tab={}
for ln in open('gpurun_out/r05aa/dw_nx.log'):
    if ln.startswith('==='): cur=ln.split()[1]; continue
    if ln.startswith(('dw3x3','dense')): shape=ln.strip(); continue
    m=re.match(r'\s+(.+?)\s+([\d.]+) us\s+([\d.]+) TB/s',ln)
    if m: tab.setdefault((shape,m.group(1)),{})[cur]=(float(m.group(2)),float(m.group(3)))
vs=["stock","dw_nx1"]
print(f"{'':70s}"+''.join(f"{v:>18s}" for v in vs))
for (sh,w),d in tab.items(): print(f"{sh[:44]:44s} {w[:24]:24s} "+''.join(f"{d.get(v,(0,0))[0]:9.1f}{d.get(v,(0,0))[1]:8.2f}T" for v in vs))
PY
unset TSII_LIBRARY
timeout 400 python bench.py --model XceptionTextSegment --size 1024 --batch 8 --storage bf16 --steps 10 --warmup 3 2>/dev/null | tail -1 > gpurun_out/r05aa/bench_cfg5_bf16.json
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r05aa/bench_cfg5_bf16.json').read())
print(d['value'], d['ms_per_step'], d['forward_only']['ms_per_step'], {k:v['ms_per_step'] for k,v in d['kernel_classes'].items()})
PY
