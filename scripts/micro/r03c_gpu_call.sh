#!/bin/bash
mkdir -p gpurun_out
(timeout 900 python bench.py --steps 20 --warmup 0 --no-cpu-baseline --no-saturated 2>gpurun_out/r03c_bench.err | tail -1) > gpurun_out/r03c_bench.json
(timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_twin.py tests/test_gpu_boundary.py -q --tb=short -p no:cacheprovider --timeout 600 2>&1 | tail -25) > gpurun_out/r03c_pytest.log
(timeout 600 python bench.py --config 4 --steps 4 --warmup 0 --no-cpu-baseline 2>>gpurun_out/r03c_bench.err | tail -1) > gpurun_out/r03c_bench4.json
cut -c1-200 gpurun_out/r03c_bench.json; tail -8 gpurun_out/r03c_pytest.log; python -c "
import json
d=json.loads([l for l in open('gpurun_out/r03c_bench4.json') if l.startswith('{')][0]); print(d['value'], d['roofline']['frac'], d['roofline']['traffic'], d['roofline']['traffic_note'][:120])"
