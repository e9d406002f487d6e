cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 600 python bench.py --steps 8 --warmup 2 --no-cpu-baseline 2> gpurun_out/bench26.err | tail -1 > gpurun_out/bench26.json; cut -c1-250 gpurun_out/bench26.json; python -c "
import json; d=json.load(open('gpurun_out/bench26.json')); print(d['launch'], d['eager_ms_per_step'], d['ms_per_step'], d['roofline']['achieved'])"; grep -v Warning gpurun_out/bench26.err | tail -5
