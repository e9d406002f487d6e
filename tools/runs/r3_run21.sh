#!/bin/bash
# round 3, run 21: same-wave MFMA + VALU interleave probe; segmentation-recipe gradient diagnostic; the two failing tests again
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; mkdir -p gpurun_out
timeout 240 tools/probes/_bin/mfma_valu_mix 4000 > gpurun_out/r03q_mfma_valu_mix.log 2>&1; echo "mix rc=$?"; cat gpurun_out/r03q_mfma_valu_mix.log
timeout 600 python tests/diag/seg_grad_probe.py > gpurun_out/r03q_seg_grad_probe.log 2>&1; echo "probe rc=$?"; grep -v amdgpu.ids gpurun_out/r03q_seg_grad_probe.log | tail -30
timeout 900 python -m pytest tests/test_parity_r2.py tests/test_training_recipe.py -m gpu -q -s -k "mixed_bf16 or recipe" > gpurun_out/r03q_gputests.log 2>&1; echo "gputests rc=$?"; grep -n "^FAILED\|^ERROR\|passed\|failed\|mixed bf16\]" gpurun_out/r03q_gputests.log | tail
