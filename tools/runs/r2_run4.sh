#!/bin/bash
# round-2 GPU run 4: ablations of the split NT kernel + the noise-aware 256^2 seg-net test with its ratio print-out
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; mkdir -p gpurun_out
for abl in 0 1 2 8 16 3 11 27; do
echo "== abl $abl"
TSII_GEMM_ABL=$abl timeout 300 python tools/gemm_bench.py --iters 5 --only fwd --modes 6 --shapes 1,3,6,9,13 2>&1 | grep mode= | cut -c1-140
done > gpurun_out/r02d_ablation.log 2>&1
cat gpurun_out/r02d_ablation.log
timeout 900 python -m pytest tests/test_parity_r2.py tests/test_memory_savers.py -m gpu -q -k "seg_nets_256 or checkpoint" > gpurun_out/r02d_seg256.log 2>&1; echo "seg256 rc=$?"; grep -E "256 grads|memory\]|passed|failed|Error" gpurun_out/r02d_seg256.log | head
