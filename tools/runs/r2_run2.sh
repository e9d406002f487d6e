#!/bin/bash
# round-2 GPU run 2: parity suite, GEMM microbench (NT + TN) over modes, SQ counters of the split kernels, bench line
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r02b_gputests.log 2>&1; echo "gpu tests rc=$?"; tail -3 gpurun_out/r02b_gputests.log
timeout 600 python tools/gemm_bench.py --iters 5 --modes 0,6,3 > gpurun_out/r02b_gemm.log 2>&1; echo "gemm rc=$?"
cut -c1-200 gpurun_out/r02b_gemm.log
TSII_GEMM_PRODUCTS=6 timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r02b_bench_mode6.log 2>&1; echo "bench rc=$?"; tail -1 gpurun_out/r02b_bench_mode6.log | cut -c1-300
export TMPDIR=/tmp; cd /tmp
rocprofv3 -L > $R/gpurun_out/r02b_counters_list.txt 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $R/gpurun_out/r02b_pmc_sq -o pmc --output-format csv -- python $R/tools/gemm_bench.py --iters 2 --modes 6 --shapes 1,6,9,13 > $R/gpurun_out/r02b_pmc_sq.log 2>&1; echo "pmc sq rc=$?"
timeout 600 rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INST_CYCLES_VMEM -d $R/gpurun_out/r02b_pmc_sq2 -o pmc --output-format csv -- python $R/tools/gemm_bench.py --iters 2 --modes 6 --shapes 1,6,9,13 > $R/gpurun_out/r02b_pmc_sq2.log 2>&1; echo "pmc sq2 rc=$?"
cd $R
python tools/pmc_summary.py sq gpurun_out/r02b_pmc_sq/pmc_counter_collection.csv gpurun_out/r02b_pmc_sq_gemm.csv "gemm_bench shapes 1,6,9,13 mode 6" ; cat gpurun_out/r02b_pmc_sq_gemm.csv | cut -c1-250 | head
find gpurun_out/r02b_pmc_sq2 -name "*counter_collection.csv" | head -1 | xargs -I{} python - {} <<'PY'
import csv, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for row in csv.DictReader(open(sys.argv[1])):
    agg[row["Kernel_Name"][:60]][row["Counter_Name"]] += float(row["Counter_Value"])
for k, v in agg.items():
    if "gemm" in k: print(k, dict(v))
PY
rm -rf gpurun_out/r02b_pmc_sq gpurun_out/r02b_pmc_sq2
