#!/bin/bash
# round 4, run 32: SQ counters (issue / wait / MFMA busy / LDS conflicts) at the final sources
set -u
ulimit -c 0
R=${GRAFT_REPO_ROOT:-$(pwd)}; mkdir -p $R/gpurun_out
export TMPDIR=/tmp; cd /tmp
timeout 240 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $R/gpurun_out/r04_pmc_sq -o pmc --output-format csv -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-f32-leg > $R/gpurun_out/r04_pmc_sq.log 2>&1; echo "pmc sq rc=$?"
cd $R
NOTE="Each pass wraps \`python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-f32-leg\` (ImageFill 512x512, 32 imgs: 2 train steps + the 3-step per-class pass + 2 forward-only steps)."
python tools/pmc_summary.py sq gpurun_out/r04_pmc_sq/pmc_counter_collection.csv gpurun_out/r04_pmc_sq_bs32.csv "$NOTE"
rm -rf gpurun_out/r04_pmc_sq
grep -E "head_cat|dw_lean_s2|dw_small" gpurun_out/r04_pmc_sq_bs32.csv | cut -c1-200
