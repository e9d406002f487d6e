#!/bin/bash
# round 5, run ai: HBM traffic (FETCH_SIZE / WRITE_SIZE passes) of the cfg 5 line in bf16 storage -> profiles/r05h_pmc_hbm_traffic_cfg5_bf16storage.{csv,json}
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R"; mkdir -p gpurun_out/r05ai
export TMPDIR=/tmp; cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 400 rocprofv3 --kernel-trace --pmc $c -d $R/gpurun_out/r05ai/pmc_$c -o pmc --output-format csv -- python $R/bench.py --model XceptionTextSegment --size 1024 --batch 8 --storage bf16 --steps 1 --warmup 1 --no-cpu-baseline > $R/gpurun_out/r05ai/pmc_$c.log 2>&1; echo "pmc $c rc=$?"
done
cd $R
PMC_SUMMARY_ALL_FILES=1 python tools/pmc_summary.py hbm $(find gpurun_out/r05ai/pmc_FETCH_SIZE -name pmc_counter_collection.csv | head -1) $(find gpurun_out/r05ai/pmc_WRITE_SIZE -name pmc_counter_collection.csv | head -1) gpurun_out/r05ai/r05h_pmc_hbm_traffic_cfg5_bf16storage "Each pass wraps python bench.py --model XceptionTextSegment --size 1024 --batch 8 --storage bf16 --steps 1 --warmup 1 --no-cpu-baseline (2 train steps + the 3-step per-class pass + 2 forward-only steps)."
rm -rf gpurun_out/r05ai/pmc_FETCH_SIZE gpurun_out/r05ai/pmc_WRITE_SIZE
head -12 gpurun_out/r05ai/r05h_pmc_hbm_traffic_cfg5_bf16storage.csv | cut -c1-200
