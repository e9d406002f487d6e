cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 600 python bench.py --model ImageFillOrigin --batch 8 --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-260
timeout 600 python tools/profile_step.py --model ImageFillOrigin --batch 8 > gpurun_out/profile_origin15.log 2>&1; head -12 gpurun_out/profile_origin15.log | tail -11; sed -n '/per shape/,$p' gpurun_out/profile_origin15.log | head -12 | cut -c1-150
