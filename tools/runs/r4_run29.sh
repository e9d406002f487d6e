#!/bin/bash
# round 4, run 29: host pipeline: resident vs fed (float32 items) vs fed (compact uint8 items expanded on the GPU)
set -u
ulimit -c 0
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; mkdir -p gpurun_out
timeout 900 python tools/host_pipeline.py --workers 1,8,16,32 --steps 12 2>&1 | tail -1 > gpurun_out/r04v_host_pipeline.log; cat gpurun_out/r04v_host_pipeline.log
timeout 300 python -m pytest tests/test_dataloader.py -m gpu -q 2>&1 | tail -1
