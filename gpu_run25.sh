cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q -k "graph_replay" > gpurun_out/pytest_gpu25.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/pytest_gpu25.log
timeout 600 python bench.py --steps 8 --warmup 2 --no-cpu-baseline 2> gpurun_out/bench25.err | tail -1 > gpurun_out/bench25.json; cat gpurun_out/bench25.json | cut -c1-250; python -c "
import json; d=json.load(open('gpurun_out/bench25.json')); print(d['launch'], d['eager_ms_per_step'], d['ms_per_step'], d['roofline']['achieved'])"; tail -3 gpurun_out/bench25.err
