#!/bin/bash
# round 5, run n: the counter-phase 256 x 256 kernel (stock) against the lock-step forms (dl0: 8 waves, dl1: 4 waves x 2 blocks) and the 128-row kernel (nodl)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R"; mkdir -p gpurun_out/r05n
timeout 300 python -m pytest tests/test_bf16_kernels.py -m gpu -q -x -k "pointwise or dense" 2>&1 | tail -3
for v in stock dl0 dl1 nodl; do
  if [ $v = stock ]; then unset TSII_LIBRARY; else export TSII_LIBRARY=$R/tools/variants/_bin/libtsii_nt_$v.so; fi
  echo "=== $v"; timeout 300 python tools/bf16_bench.py --only pw --big 2>&1 | grep -v "dW\|amdgpu.ids"
done > gpurun_out/r05n/nt_pp.log 2>&1
python - <<'PY'
import re
cur=None; tab={}
for ln in open('gpurun_out/r05n/nt_pp.log'):
    if ln.startswith('==='): cur=ln.split()[1]; continue
    if ln.startswith(('1x1','dense')): shape=ln.strip(); continue
    m=re.match(r'\s+(.+?)\s+([\d.]+) us',ln)
    if m: tab.setdefault((shape,m.group(1)),{})[cur]=float(m.group(2))
vs=["stock","dl0","dl1","nodl"]
print(f"{'':70s}"+''.join(f"{v:>9s}" for v in vs))
for (sh,w),d in tab.items(): print(f"{sh[:44]:44s} {w[:24]:24s} "+''.join(f"{d.get(v,0):9.1f}" for v in vs))
PY
