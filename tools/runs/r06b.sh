#!/bin/bash
# round 6, call b: K3p tile dealing (contiguous vs per-XCD round robin) and consumer lag, every NT launch of the step replayed
mkdir -p gpurun_out/r06b
V=tools/variants/_bin/libtsii_abl.so
for opt in 0 16 48 80 32; do
  TSII_LIBRARY=$V TSII_GEMM_PC_OPT=$opt python tools/nt_bench.py --iters 10 > gpurun_out/r06b/nt_opt$opt.log 2>&1
  echo "opt=$opt: $(tail -1 gpurun_out/r06b/nt_opt$opt.log)"
done
python -m pytest tests/test_parity_r2.py -m gpu -x -q -s -k "256" > gpurun_out/r06b/tests_seg256.log 2>&1; echo "seg256 rc=$?"
python -m pytest tests/test_bf16_storage.py -m gpu -x -q -s -k "fixture" > gpurun_out/r06b/tests_bf16.log 2>&1; echo "bf16 fixture rc=$?"
grep -n "outlier" gpurun_out/r06b/tests_seg256.log
