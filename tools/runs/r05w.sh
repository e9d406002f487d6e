#!/bin/bash
# round 5, run w: kernel-only durations of the 1x1 forms on 131072 x 512 x 512 (rocprofv3 kernel stats)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R"; mkdir -p gpurun_out/r05w
export TMPDIR=/tmp; cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r05w/prof -o pw --output-format csv -- python $R/tools/bf16_bench.py --only pw --split-fused --pw-shapes "131072,512,512" > $R/gpurun_out/r05w/run.log 2>&1
cd $R
python - <<'PY'
import csv,glob
f=glob.glob('gpurun_out/r05w/prof/**/pw_kernel_stats.csv',recursive=True)[0]
rows=list(csv.DictReader(open(f)))
for r in sorted(rows,key=lambda r:-float(r['TotalDurationNs']))[:14]:
    print(f"calls {int(r['Calls']):4d} avg {float(r['AverageNs'])/1e3:8.1f} us min {float(r['MinNs'])/1e3:8.1f}  {r['Name'][:120]}")
PY
rm -rf gpurun_out/r05w/prof
