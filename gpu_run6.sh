cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu6.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/pytest_gpu6.log
timeout 600 python tools/profile_step.py > gpurun_out/profile_step6.log 2>&1; echo "profile rc=$?"; head -14 gpurun_out/profile_step6.log | tail -13
timeout 600 python bench.py --steps 8 --warmup 2 --no-cpu-baseline > gpurun_out/bench6_b32.log 2>&1; tail -1 gpurun_out/bench6_b32.log | cut -c1-330
export TMPDIR=/tmp; cd /tmp; R=$GRAFT_REPO_ROOT
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/gpurun_out/pmc6_fetch -o fetch --output-format csv -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $R/gpurun_out/pmc6_fetch.log 2>&1; echo "pmc fetch rc=$?"
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $R/gpurun_out/pmc6_write -o write --output-format csv -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $R/gpurun_out/pmc6_write.log 2>&1; echo "pmc write rc=$?"
