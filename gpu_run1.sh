cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
python __graft_entry__.py smoke > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?"; tail -3 gpurun_out/smoke.log
timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -15 gpurun_out/pytest_gpu.log
timeout 600 python bench.py --steps 5 --warmup 2 --batch 8 --no-cpu-baseline > gpurun_out/bench_b8.log 2>&1; echo "bench8 rc=$?"; tail -2 gpurun_out/bench_b8.log
timeout 900 python bench.py --steps 5 --warmup 2 > gpurun_out/bench_b32.log 2>&1; echo "bench32 rc=$?"; tail -2 gpurun_out/bench_b32.log
export TMPDIR=/tmp; cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_r1 -o b32 -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/rocprof.log 2>&1; echo "rocprof rc=$?"
cd $GRAFT_REPO_ROOT; ls -R gpurun_out/prof_r1 | head -20; rocm-smi --showmeminfo vram 2>/dev/null | head -5
