#!/bin/bash
# round 5, run ah: SQ counters (matrix-pipe busy share, waits, LDS conflicts) of the bf16 GEMM kernels under the entry-point microbench
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R"; mkdir -p gpurun_out/r05ah
export TMPDIR=/tmp; cd /tmp
timeout 400 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $R/gpurun_out/r05ah/pmc -o pmc --output-format csv -- python $R/tools/bf16_bench.py --only pw --iters 3 --pw-shapes "65536,4096,4096;131072,512,512" > $R/gpurun_out/r05ah/run.log 2>&1; echo "rc=$?"
cd $R
python tools/pmc_summary.py sq $(find gpurun_out/r05ah/pmc -name "pmc_counter_collection.csv" | head -1) gpurun_out/r05ah/pmc_sq_bf16_gemm.csv "rocprofv3 --pmc SQ_* around tools/bf16_bench.py --only pw --iters 3 --pw-shapes 65536,4096,4096;131072,512,512 (plain / fused forward, dX, dW forms of both shapes mixed per kernel)"
rm -rf gpurun_out/r05ah/pmc
head -30 gpurun_out/r05ah/pmc_sq_bf16_gemm.csv | cut -c1-260
