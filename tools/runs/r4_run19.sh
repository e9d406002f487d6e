#!/bin/bash
# round 4, run 19: head dW kernel: 4 vs 3 waves, ablations (neither half / no products)
set -u
ulimit -c 0
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; mkdir -p gpurun_out
python tools/head_dw_bench.py 2>&1 | tail -1
for v in hm3 hma3 hma4; do TSII_LIBRARY=$R/tools/variants/_bin/libtsii_$v.so python tools/head_dw_bench.py 2>&1 | tail -1; done
