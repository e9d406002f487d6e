cd $GRAFT_REPO_ROOT
timeout 600 python tools/gemm_bench.py 2>&1 | grep "^M="
