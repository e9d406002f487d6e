cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu33.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/pytest_gpu33.log
timeout 600 python tools/profile_step.py > gpurun_out/profile_step33.log 2>&1; grep -E "^  tsii_|total" gpurun_out/profile_step33.log | cut -c1-60 | head -12
timeout 600 python bench.py --steps 8 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['forward_only'], d['roofline']['achieved'])"
