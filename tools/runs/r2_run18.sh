#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_training_recipe.py -m gpu -q -s > gpurun_out/r02p_recipe_tests.log 2>&1; echo "rc=$?"; tail -25 gpurun_out/r02p_recipe_tests.log
