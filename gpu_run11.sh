cd $GRAFT_REPO_ROOT
for v in 0 1 2 3 4 9; do echo "== TSII_DW_TILE=$v"; TSII_DW_TILE=$v timeout 300 python tools/dw_bench.py 2>&1 | grep "^dw"; done
