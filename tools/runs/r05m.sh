#!/bin/bash
# round 5, run m: the 256 x 256 direct-to-LDS NT kernel against the 128-row kernel (HNT_DL=0) and its 128 x 256 form (HNT_WIDE)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R"; mkdir -p gpurun_out/r05m
timeout 300 python -m pytest tests/test_bf16_kernels.py -m gpu -q -x -k "pointwise or dense" 2>&1 | tail -3
for v in stock nodl wide; do
  if [ $v = stock ]; then unset TSII_LIBRARY; else export TSII_LIBRARY=$R/tools/variants/_bin/libtsii_nt_$v.so; fi
  echo "=== $v"; timeout 300 python tools/bf16_bench.py --only pw,dense --big 2>&1 | grep -v "dW\|amdgpu.ids"
done > gpurun_out/r05m/nt_dl.log 2>&1
python - <<'PY'
import re
cur=None; tab={}
for ln in open('gpurun_out/r05m/nt_dl.log'):
    if ln.startswith('==='): cur=ln.split()[1]; continue
    if ln.startswith(('1x1','dense')): shape=ln.strip(); continue
    m=re.match(r'\s+(.+?)\s+([\d.]+) us',ln)
    if m: tab.setdefault((shape,m.group(1)),{})[cur]=float(m.group(2))
vs=["stock","nodl","wide"]
print(f"{'':70s}"+''.join(f"{v:>9s}" for v in vs))
for (sh,w),d in tab.items(): print(f"{sh[:44]:44s} {w[:24]:24s} "+''.join(f"{d.get(v,0):9.1f}" for v in vs))
PY
