cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu39.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/pytest_gpu39.log
timeout 600 python bench.py --steps 8 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['forward_only']['value'], d['roofline']['achieved'], d['roofline']['frac'], d['roofline']['traffic'])"
timeout 600 python bench.py --model ImageFillOrigin --batch 32 --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-160
timeout 600 python bench.py --model ImageFillOriginV2 --batch 32 --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-160
timeout 300 python tools/seg_step.py --model TextSegament --batch 8 2>&1 | tail -1
timeout 300 python tools/seg_step.py --model XceptionTextSegment --batch 8 2>&1 | tail -1
