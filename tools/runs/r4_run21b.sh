#!/bin/bash
set -u
ulimit -c 0
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; mkdir -p gpurun_out
python tools/head_dw_bench.py 2>&1 | grep forward
for v in hfa1 hfa2 hfa4; do TSII_LIBRARY=$R/tools/variants/_bin/libtsii_$v.so python tools/head_dw_bench.py 2>&1 | grep forward; done
