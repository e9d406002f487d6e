#!/bin/bash
# round 3, run 41: bench-line contract test (ImageFill 128x128 through bench.py) at the final state
set -u
ulimit -c 0
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; mkdir -p gpurun_out
timeout 100 python -m pytest tests/test_training_recipe.py -m gpu -q -k "contract and extra2" 2>&1 | tail -3
