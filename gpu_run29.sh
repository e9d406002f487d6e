cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 600 python tools/profile_step.py > gpurun_out/profile_step29.log 2>&1; grep -E "^  tsii_pw|total" gpurun_out/profile_step29.log | cut -c1-60 | head -8; grep -E "^tsii_pw" gpurun_out/profile_step29.log | cut -c1-110 | head -12
