cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; R=$GRAFT_REPO_ROOT
export TMPDIR=/tmp; cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof34 -o b32 --output-format csv -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $R/gpurun_out/rocprof34.log 2>&1; echo "rocprof stats rc=$?"
cp $R/gpurun_out/prof34/b32_kernel_stats.csv $R/gpurun_out/kernel_stats34.csv
