#!/bin/bash
# round 4, run 20: wider row reduction / colsum, head dW with prefetch: parity tests that touch them, bench, small-kernel stats
set -u
ulimit -c 0
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_parity_ops.py -m gpu -q -x > gpurun_out/r04q_tests.log 2>&1; tail -3 gpurun_out/r04q_tests.log
for i in 1 2; do timeout 300 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-f32-leg 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('bench', d['ms_per_step'], d['value'], d.get('forward_only'))"; done
cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/prof20; timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof20 -o r20 --output-format csv -- python $R/bench.py --steps 6 --warmup 4 --no-cpu-baseline --no-f32-leg > $R/gpurun_out/r04q_rocprof.log 2>&1; echo rc=$?
f=$(find /tmp/prof20 -name "*kernel_stats.csv" | head -1); cp "$f" $R/gpurun_out/r04q_kernel_stats.csv
python - <<PY
import csv
rows = list(csv.DictReader(open("$R/gpurun_out/r04q_kernel_stats.csv")))
for r in rows:
    n = r["Name"]
    if any(k in n for k in ("head_", "colsum", "reduce_rows", "gemm_tn_kernel")):
        print(n[:60], r["Calls"], r["AverageNs"], r["MinNs"], r["MaxNs"])
PY
