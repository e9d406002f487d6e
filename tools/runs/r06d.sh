#!/bin/bash
# round 6, call d: bf16 load-time BatchNorm fast paths (A/B builds), K3p dealt per XCD as the stock default: parity + bench
mkdir -p gpurun_out/r06d
for v in stock dwold dwlb3 gemmold; do
  if [ $v = stock ]; then unset TSII_LIBRARY; else export TSII_LIBRARY=tools/variants/_bin/libtsii_$v.so; fi
  python tools/bf16_bench.py --only dw,pw --iters 20 > gpurun_out/r06d/bf16_bench_$v.log 2>&1
done
unset TSII_LIBRARY
python bench.py --model XceptionTextSegment --size 1024 --batch 8 --storage bf16 --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/r06d/bench_cfg5.json
python -m pytest tests/test_parity_ops.py tests/test_workload_sizes.py tests/test_decoder_split_resolution.py -m gpu -x -q > gpurun_out/r06d/tests_parity.log 2>&1; echo "parity rc=$?"
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary 2>&1 | tail -1 > gpurun_out/r06d/bench_bs32.json
python - <<'PY'
import json
for f in ("bench_cfg5","bench_bs32"):
    d=json.load(open(f"gpurun_out/r06d/{f}.json")); print(f, d["value"], d["ms_per_step"], d["forward_only"]["ms_per_step"], {k:v["ms_per_step"] for k,v in d["kernel_classes"].items()})
PY
tail -3 gpurun_out/r06d/tests_parity.log
