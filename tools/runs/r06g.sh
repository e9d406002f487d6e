#!/bin/bash
# round 6, call g: 128 x 256 tiles for thin reductions (A/B + on-chip check), then the round's collection (tools/collect_profiles.sh r06)
mkdir -p gpurun_out/r06g
V=tools/variants/_bin/libtsii_abl.so
for wk in 0 128; do
  TSII_LIBRARY=$V TSII_GEMM_PC_WIDE_K=$wk python tools/nt_bench.py --iters 10 --max-k 128 > gpurun_out/r06g/nt_widek$wk.log 2>&1
  echo "wide_k=$wk: $(tail -1 gpurun_out/r06g/nt_widek$wk.log)"
done
TSII_LIBRARY=$V TSII_GEMM_PC_WIDE_K=128 python tools/pc_check.py --runs 2 > gpurun_out/r06g/pc_check_widek128.log 2>&1; echo "pc_check rc=$?"; tail -3 gpurun_out/r06g/pc_check_widek128.log
bash tools/collect_profiles.sh r06 2>&1 | tail -40
