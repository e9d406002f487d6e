cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu2.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/pytest_gpu2.log
timeout 600 python tools/profile_step.py > gpurun_out/profile_step2.log 2>&1; echo "profile rc=$?"; head -40 gpurun_out/profile_step2.log
timeout 900 python bench.py --steps 8 --warmup 2 > gpurun_out/bench2_b32.log 2>&1; echo "bench32 rc=$?"; tail -1 gpurun_out/bench2_b32.log
export TMPDIR=/tmp; cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_r2 -o b32 -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/rocprof2.log 2>&1; echo "rocprof rc=$?"
