cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 600 python tools/profile_step.py > gpurun_out/profile_step14.log 2>&1; head -16 gpurun_out/profile_step14.log | tail -15
timeout 600 python bench.py --steps 8 --warmup 2 --no-cpu-baseline > gpurun_out/bench14_b32.log 2>&1; tail -1 gpurun_out/bench14_b32.log | cut -c1-1100
