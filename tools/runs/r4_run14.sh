#!/bin/bash
# round 4, run 14: lean strip kernel vs the round-3 strip kernel on the exact depth-wise calls of TextSegament 256^2 (and ImageFill 256^2)
set -u
ulimit -c 0
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; mkdir -p gpurun_out
timeout 300 python tools/dw_ab.py --other tools/variants/_bin/libtsii_nolean.so --model TextSegament --size 256 --batch 2 > gpurun_out/r04l_dw_ab_textsegament.log 2>&1; tail -60 gpurun_out/r04l_dw_ab_textsegament.log
