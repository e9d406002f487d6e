cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python bench.py --steps 10 --warmup 3 2>&1 | tail -1 > gpurun_out/bench37.json; python -c "
import json; d=json.load(open('gpurun_out/bench37.json')); print(d['value'], d['ms_per_step'], d['forward_only'], d['roofline']['achieved'], d['roofline']['frac'], d['cpu_baseline']['value'])"
timeout 300 python tools/seg_step.py --model TextSegament --batch 8 2>&1 | tail -1
timeout 600 python bench.py --model ImageFillOrigin --batch 32 --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-160
