#!/bin/bash
# round 4, run 21: head forward on the matrix cores: microbench, parity, bench
set -u
ulimit -c 0
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; mkdir -p gpurun_out
python tools/head_dw_bench.py 2>&1 | tail -2
timeout 900 python -m pytest tests/test_parity_ops.py -m gpu -q -x -k "head_over_virtual or imagefill" > gpurun_out/r04r_tests.log 2>&1; tail -3 gpurun_out/r04r_tests.log
for v in 0 1; do TSII_HEAD_MFMA=$v timeout 300 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-f32-leg 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('head_mfma=$v', d['ms_per_step'], d['value'], d.get('forward_only'))"; done
