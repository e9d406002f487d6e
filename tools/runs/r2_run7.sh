#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/r02g_gputests.log 2>&1; echo "gpu tests rc=$?"; grep -E "median|passed|failed|^FAILED|^E  " gpurun_out/r02g_gputests.log | cut -c1-220 | head -30
for k in 0 1; do
TSII_DW_OCC3=$k timeout 600 python bench.py --steps 15 --warmup 4 --no-cpu-baseline --no-f32-leg > gpurun_out/r02g_bench_occ$k.log 2>&1; echo "bench occ3=$k rc=$?"; tail -1 gpurun_out/r02g_bench_occ$k.log | python -c "
import sys, json
j = json.loads(sys.stdin.read()); print(j['value'], j['ms_per_step'], j['forward_only']['ms_per_step']); [print('   ', c, v['ms_per_step'], v.get('tb_per_s')) for c, v in j['kernel_classes'].items()]"
done
TSII_DW_OCC3=1 timeout 600 python tools/profile_step.py > gpurun_out/r02g_per_shape_occ1.log 2>&1; grep -E "dw_fwd_bn|dense_bwd_dw|dense_fwd" gpurun_out/r02g_per_shape_occ1.log | head -12
timeout 300 python tools/seg_step.py --model TextSegament --batch 32 --size 512 > gpurun_out/r02g_seg.log 2>&1
timeout 300 python tools/seg_step.py --model TextSegament --batch 64 --size 512 --checkpoint >> gpurun_out/r02g_seg.log 2>&1
timeout 300 python tools/seg_step.py --model TextSegament --batch 64 --size 512 --checkpoint --pixel-shuffle >> gpurun_out/r02g_seg.log 2>&1
timeout 300 python tools/seg_step.py --model TextSegament --batch 64 --size 512 >> gpurun_out/r02g_seg.log 2>&1
timeout 300 python tools/seg_step.py --model XceptionTextSegment --batch 8 --size 1024 >> gpurun_out/r02g_seg.log 2>&1
timeout 300 python tools/seg_step.py --model XceptionTextSegment --batch 8 --size 1024 --products 1 >> gpurun_out/r02g_seg.log 2>&1
timeout 300 python tools/seg_step.py --model XceptionTextSegment --batch 8 --size 1024 --products 0 >> gpurun_out/r02g_seg.log 2>&1
grep -v amdgpu gpurun_out/r02g_seg.log
