#!/bin/bash
# round 3, run 6: which part of the producers slows the consumers? (ablations + clock from GRBM_GUI_ACTIVE)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; mkdir -p gpurun_out
export TMPDIR=/tmp
for a in 0 400 432 464 496 96 128; do
  TSII_GEMM_PC_ABL=$a timeout 120 python tools/pc_probe.py 65536 1024 1024 10 2>&1 | grep -v amdgpu.ids | sed -e "s/^/abl=$a /"
done
for o in 12 1 4; do
  TSII_GEMM_PC_OPT=$o timeout 120 python tools/pc_probe.py 65536 1024 1024 10 2>&1 | grep -v amdgpu.ids | sed -e "s/^/opt=$o /"
done
TSII_GEMM_PC=0 timeout 120 python tools/pc_probe.py 65536 1024 1024 10 2>&1 | grep -v amdgpu.ids
cd /tmp
for a in 0 400 464; do
TSII_GEMM_PC_ABL=$a timeout 300 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU -d $R/gpurun_out/r03f_pmc$a -o pmc --output-format csv -- python $R/tools/pc_probe.py 65536 1024 1024 3 > $R/gpurun_out/r03f_pmc$a.log 2>&1; echo "pmc abl=$a rc=$?"
done
cd $R
python - <<'PY'
import csv, glob, collections
for d in sorted(glob.glob("gpurun_out/r03f_pmc*/")):
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"][:70]
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(k, r["Counter_Name"])] += 1
        for k, v in acc.items():
            if "gemm_nt" in k: print(d, k, {c: x / cnt[(k, c)] for c, x in v.items()})
    for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
        rows = [r for r in csv.DictReader(open(f)) if "gemm_nt" in r["Kernel_Name"]]
        ds = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows]
        print(d, "durations us: last 3", ds[-3:])
PY
rm -rf gpurun_out/r03f_pmc*/
