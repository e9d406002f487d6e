#!/bin/bash
# round 3, run 3: ablations + PMC of the producer / consumer GEMM on one compute-bound shape
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; mkdir -p gpurun_out
export TMPDIR=/tmp
for o in 12 28 44 76 92 140 220; do
  TSII_GEMM_PC_OPT=$o timeout 120 python tools/pc_probe.py 65536 1024 1024 2>&1 | grep -v amdgpu.ids
done
for o in 12 76 92; do
  TSII_GEMM_PC_OPT=$o timeout 120 python tools/pc_probe.py 2097152 192 384 2>&1 | grep -v amdgpu.ids
done
TSII_GEMM_PC=0 timeout 120 python tools/pc_probe.py 65536 1024 1024 2>&1 | grep -v amdgpu.ids
cd /tmp
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $R/gpurun_out/r03c_pmc1 -o pmc --output-format csv -- python $R/tools/pc_probe.py 65536 1024 1024 3 > $R/gpurun_out/r03c_pmc1.log 2>&1; echo "pmc1 rc=$?"
timeout 300 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_LEVEL_LDS SQ_LDS_DATA_FIFO_FULL -d $R/gpurun_out/r03c_pmc2 -o pmc --output-format csv -- python $R/tools/pc_probe.py 65536 1024 1024 3 > $R/gpurun_out/r03c_pmc2.log 2>&1; echo "pmc2 rc=$?"
cd $R
python - <<'PY'
import csv, glob, collections
for d in ("gpurun_out/r03c_pmc1", "gpurun_out/r03c_pmc2"):
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"][:60]
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        for k, v in acc.items():
            if "gemm" in k or "split" in k: print(f, k, dict(v))
    for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
        rows = [r for r in csv.DictReader(open(f)) if "gemm_nt" in r["Kernel_Name"]]
        print(f, [ (int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))/1e3 for r in rows], rows[0]["VGPR_Count"] if rows else None, rows[0].get("Scratch_Size") if rows else None, rows[0].get("LDS_Block_Size") if rows else None)
PY
rm -rf gpurun_out/r03c_pmc1 gpurun_out/r03c_pmc2
