#!/bin/bash
# round 5, run p: the quadrant-phase 256 x 256 x 64 kernel (stock) against the K-slab counter-phase kernel (pp) and the lock-step one (dl0)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R"; mkdir -p gpurun_out/r05q
timeout 300 python -m pytest tests/test_bf16_kernels.py -m gpu -q -x -k "pointwise or dense" 2>&1 | tail -3
for v in stock phL phbnb; do
  if [ $v = stock ]; then unset TSII_LIBRARY; else export TSII_LIBRARY=$R/tools/variants/_bin/libtsii_nt_$v.so; fi
  echo "=== $v"; timeout 300 python tools/bf16_bench.py --only pw --big 2>&1 | grep -v "dW\|amdgpu.ids\|BN-on-load"
done > gpurun_out/r05q/nt_ph.log 2>&1
python - <<'PY'
import re
cur=None; tab={}
for ln in open('gpurun_out/r05q/nt_ph.log'):
    if ln.startswith('==='): cur=ln.split()[1]; continue
    if ln.startswith(('1x1','dense')): shape=ln.strip(); continue
    m=re.match(r'\s+(.+?)\s+([\d.]+) us.*?([\d.]+) TF/s',ln)
    if m: tab.setdefault((shape,m.group(1)),{})[cur]=(float(m.group(2)),float(m.group(3)))
vs=["stock","phL","phbnb"]
print(f"{'':70s}"+''.join(f"{v:>18s}" for v in vs))
for (sh,w),d in tab.items(): print(f"{sh[:44]:44s} {w[:24]:24s} "+''.join(f"{d.get(v,(0,0))[0]:9.1f}{d.get(v,(0,0))[1]:8.0f}T" for v in vs))
PY
