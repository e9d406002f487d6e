#!/bin/bash
# round 3, run 11: 912 = MFMA-only consumers + A loads; 9104 = the same, loads always hit cache; 17296 = the same with 2 stages in flight; + VMEM counters
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; mkdir -p gpurun_out
export TMPDIR=/tmp
for a in 912 9104 17296 464; do
  TSII_GEMM_PC_ABL=$a timeout 120 python tools/pc_probe.py 65536 1024 1024 10 2>&1 | grep -v amdgpu.ids | sed -e "s/^/abl=$a /"
done
cd /tmp
for a in 912 464; do
TSII_GEMM_PC_ABL=$a timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VMEM SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_VMEM SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_INST_CYCLES_SALU SQ_VALU_MFMA_COEXEC_CYCLES GRBM_TA_BUSY GRBM_GUI_ACTIVE -d $R/gpurun_out/r03h_pmc$a -o pmc --output-format csv -- python $R/tools/pc_probe.py 65536 1024 1024 3 > $R/gpurun_out/r03h_pmc$a.log 2>&1; echo "pmc abl=$a rc=$?"
done
cd $R
python - <<'PY'
import csv, glob, collections
for d in sorted(glob.glob("gpurun_out/r03h_pmc*/")):
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"][:70]
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(k, r["Counter_Name"])] += 1
        for k, v in acc.items():
            if "gemm_nt" in k: print(d, {c: x / cnt[(k, c)] for c, x in v.items()})
PY
rm -rf gpurun_out/r03h_pmc*/
