cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 600 python tools/profile_step.py > gpurun_out/profile_step38.log 2>&1; grep -E "^  tsii_pw|total" gpurun_out/profile_step38.log | cut -c1-60 | head -8; grep -E "^tsii_pw_(fwd_bn|bwd_dx)" gpurun_out/profile_step38.log | cut -c1-110 | head -10
timeout 600 python bench.py --steps 8 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['achieved'], d['roofline']['frac'])"
