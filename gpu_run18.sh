cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu18.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/pytest_gpu18.log
timeout 600 python tools/profile_step.py > gpurun_out/profile_step18.log 2>&1; head -16 gpurun_out/profile_step18.log | tail -15
grep -E "tsii_dense" gpurun_out/profile_step18.log | tail -7 | cut -c1-140
timeout 600 python bench.py --steps 8 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-200
timeout 600 python bench.py --model ImageFillOrigin --batch 32 --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-200
timeout 600 python tools/seg_step.py --model TextSegament --batch 8 --size 512 2>&1 | grep -v "check point\|re-trained\|amdgpu.ids"
timeout 600 python tools/seg_step.py --model XceptionTextSegment --batch 8 --size 512 2>&1 | grep -v "amdgpu.ids"
