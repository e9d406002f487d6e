#!/bin/bash
set -u
ulimit -c 0
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R"; mkdir -p gpurun_out
timeout 900 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-f32-leg 2>&1 | tail -1 > gpurun_out/r05h_bench_bs32.log; python - <<'PY'
import json
d=json.loads(open('gpurun_out/r05h_bench_bs32.log').read())
print(d['value'], d['ms_per_step'], d['forward_only'])
for k,v in d['kernel_classes'].items(): print(' ',k, v['ms_per_step'], v['tb_per_s'], v.get('per_launch_roofline_frac'))
PY
timeout 1800 python -m pytest tests -m gpu -q -x > gpurun_out/r05h_gputests.log 2>&1; tail -3 gpurun_out/r05h_gputests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3
