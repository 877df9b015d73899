#!/bin/bash
mkdir -p gpurun_out
(timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 2>gpurun_out/r03h_bench.err | tail -1) > gpurun_out/r03h_bench.json
(timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2) > gpurun_out/r03h_smoke.log
(timeout 1800 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider --timeout 600 2>&1 | tail -12) > gpurun_out/r03h_pytest.log
(timeout 600 python bench.py --config 3 --steps 20 --warmup 0 --no-cpu-baseline --no-saturated 2>>gpurun_out/r03h_bench.err | tail -1) > gpurun_out/r03h_bench3.json
(timeout 600 python bench.py --config 4 --steps 4 --warmup 0 --no-cpu-baseline 2>>gpurun_out/r03h_bench.err | tail -1) > gpurun_out/r03h_bench4.json
cut -c1-260 gpurun_out/r03h_bench.json; cat gpurun_out/r03h_smoke.log; tail -4 gpurun_out/r03h_pytest.log
python -c "
import json
for f in ['gpurun_out/r03h_bench3.json','gpurun_out/r03h_bench4.json']:
    d=json.loads([l for l in open(f) if l.startswith('{')][0]); print(f, d['value'], d['seconds'], d['roofline']['frac'])"
