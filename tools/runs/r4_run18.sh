#!/bin/bash
# round 4, run 18: kernel-level times of the head / stem kernels inside the step
set -u
ulimit -c 0
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd /tmp; export TMPDIR=/tmp; mkdir -p $R/gpurun_out
rm -rf /tmp/prof18; timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof18 -o r18 --output-format csv -- python $R/bench.py --steps 6 --warmup 4 --no-cpu-baseline --no-f32-leg > $R/gpurun_out/r04p_rocprof.log 2>&1; echo rc=$?
f=$(find /tmp/prof18 -name "*kernel_stats.csv" | head -1); cp "$f" $R/gpurun_out/r04p_kernel_stats.csv
python - <<PY
import csv
rows = list(csv.DictReader(open("$R/gpurun_out/r04p_kernel_stats.csv")))
for r in rows:
    n = r["Name"]
    if any(k in n for k in ("head_", "colsum", "reduce_rows", "gemm_tn_kernel", "conv_dw_reduce", "s2d", "bn_bwd_apply", "stem")):
        print(n[:70], r["Calls"], r["AverageNs"], r["TotalDurationNs"])
PY
