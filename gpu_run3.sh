cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu3.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/pytest_gpu3.log
timeout 600 python tools/profile_step.py > gpurun_out/profile_step3.log 2>&1; echo "profile rc=$?"; head -16 gpurun_out/profile_step3.log
timeout 600 python tools/gemm_bench.py > gpurun_out/gemm_bench3.log 2>&1; echo "gemm rc=$?"; cat gpurun_out/gemm_bench3.log | tail -16
timeout 600 python bench.py --steps 8 --warmup 2 --no-cpu-baseline > gpurun_out/bench3_b32.log 2>&1; tail -1 gpurun_out/bench3_b32.log | cut -c1-400
export TMPDIR=/tmp; cd /tmp
R=$GRAFT_REPO_ROOT
timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $R/gpurun_out/pmc3_sq -o sq --output-format csv -- python $R/tools/gemm_bench.py --iters 1 --shapes 1,8,13 > $R/gpurun_out/pmc3_sq.log 2>&1; echo "pmc sq rc=$?"
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/gpurun_out/pmc3_fetch -o fetch --output-format csv -- python $R/tools/gemm_bench.py --iters 1 --shapes 1,8,13 > $R/gpurun_out/pmc3_fetch.log 2>&1; echo "pmc fetch rc=$?"
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE GRBM_GUI_ACTIVE -d $R/gpurun_out/pmc3_write -o write --output-format csv -- python $R/tools/gemm_bench.py --iters 1 --shapes 1,8,13 > $R/gpurun_out/pmc3_write.log 2>&1; echo "pmc write rc=$?"
ls -la $R/gpurun_out/pmc3_sq $R/gpurun_out/pmc3_fetch | head
