#!/bin/bash
# round 4, run 26: smoke()'s stem-dW distance from fp64 under the A/B switches of this round's kernels (is the 6.4e-4 -> 1.1e-3 move noise?)
set -u
ulimit -c 0
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; mkdir -p gpurun_out
run() { echo "== $1"; env $2 timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -E "smoke\]|Error" ; }
run "default" "A=1" > gpurun_out/r04u_smoke_ab.log
run "TSII_HEAD_MFMA=0" "TSII_HEAD_MFMA=0" >> gpurun_out/r04u_smoke_ab.log
run "TSII_FUSE_POOL_BN_BWD=0" "TSII_FUSE_POOL_BN_BWD=0" >> gpurun_out/r04u_smoke_ab.log
run "no lean stride-2 / small-map kernels" "TSII_LIBRARY=$R/tools/variants/_bin/libtsii_nol2.so" >> gpurun_out/r04u_smoke_ab.log
run "all three off" "TSII_HEAD_MFMA=0 TSII_FUSE_POOL_BN_BWD=0 TSII_LIBRARY=$R/tools/variants/_bin/libtsii_nol2.so" >> gpurun_out/r04u_smoke_ab.log
cat gpurun_out/r04u_smoke_ab.log
