#!/bin/bash
# round 6, call a: the bf16 refusal census on the chip, the seg-256 fixture tests with the partner-weight rule (prints the partner
# figures), the bf16 net test with cosines, smoke, and the default bench line with its secondary legs
mkdir -p gpurun_out/r06a
python -m pytest tests/test_bf16_refusal.py tests/test_bf16_storage.py -m gpu -x -q -s > gpurun_out/r06a/tests_bf16.log 2>&1; echo "bf16 tests rc=$?"
python -m pytest tests/test_parity_r2.py -m gpu -x -q -s -k "256" > gpurun_out/r06a/tests_seg256.log 2>&1; echo "seg256 rc=$?"
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r06a/smoke.log 2>&1; echo "smoke rc=$?"
python bench.py > gpurun_out/r06a/bench.json 2> gpurun_out/r06a/bench.err; echo "bench rc=$?"
tail -c 3000 gpurun_out/r06a/bench.json
