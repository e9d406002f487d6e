cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu9.log 2>&1; echo "pytest rc=$?"; tail -6 gpurun_out/pytest_gpu9.log
timeout 600 python examples/demo_segmentation.py --synthetic 2>&1 | grep -v "amdgpu.ids" | tail -4
timeout 600 python examples/demo_segmentation.py --synthetic --model TextSegament 2>&1 | grep -v "amdgpu.ids\|check point\|re-trained" | tail -3
timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 6 --warmup 2 --no-cpu-baseline > gpurun_out/bench9_torchrun.log 2>&1; echo "torchrun rc=$?"; tail -1 gpurun_out/bench9_torchrun.log | cut -c1-300
