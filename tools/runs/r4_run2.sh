#!/bin/bash
# round 4, run 2: lean strip with the next slab requested at the END of a step; A/B builds: non-temporal stores, 2 waves per SIMD
set -u
ulimit -c 0
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; mkdir -p gpurun_out
for v in default nt w2; do
  echo "== $v" >> gpurun_out/r04b_dw_variants.log
  if [ $v = default ]; then unset TSII_LIBRARY; else export TSII_LIBRARY=$R/tools/variants/_bin/libtsii_$v.so; fi
  timeout 120 python tools/dw_bench.py 2>&1 | grep "^dw" >> gpurun_out/r04b_dw_variants.log
done
unset TSII_LIBRARY
cat gpurun_out/r04b_dw_variants.log
export TMPDIR=/tmp; cd /tmp
timeout 200 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU -d $R/gpurun_out/r04b_pmc_sq -o pmc --output-format csv -- python $R/tools/dw_bench.py > $R/gpurun_out/r04b_pmc_sq.log 2>&1; echo "pmc rc=$?"
cd $R
python tools/pmc_summary.py sq gpurun_out/r04b_pmc_sq/pmc_counter_collection.csv gpurun_out/r04b_pmc_sq_dwbench.csv "tools/dw_bench.py" 2>&1 | tail -3
rm -rf gpurun_out/r04b_pmc_sq
cat gpurun_out/r04b_pmc_sq_dwbench.csv | head -30
