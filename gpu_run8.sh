cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu8.log 2>&1; echo "pytest rc=$?"; tail -12 gpurun_out/pytest_gpu8.log
timeout 600 python tools/seg_step.py --model TextSegament --batch 8 --size 512 2>&1 | grep -v "check point\|re-trained\|amdgpu.ids"
timeout 600 python tools/seg_step.py --model XceptionTextSegment --batch 8 --size 512 2>&1 | grep -v "amdgpu.ids"
timeout 600 python bench.py --steps 8 --warmup 2 --no-cpu-baseline > gpurun_out/bench8_b32.log 2>&1; tail -1 gpurun_out/bench8_b32.log | cut -c1-330
