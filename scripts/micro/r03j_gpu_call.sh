#!/bin/bash
mkdir -p gpurun_out
(timeout 900 python bench.py --steps 20 --warmup 0 --no-cpu-baseline --no-saturated 2>gpurun_out/r03k_bench.err | tail -1) > gpurun_out/r03k_bench.json
(timeout 900 python -m pytest tests/test_gpu_twin.py -q --tb=short -p no:cacheprovider --timeout 400 2>&1 | tail -6) > gpurun_out/r03k_twin.log
python -c "
import json
d=json.loads([l for l in open('gpurun_out/r03k_bench.json') if l.startswith('{')][0]); print(d['value'], d['seconds'], d['ess_per_sec'], d['config']['posteriors']['2016']['twin'])"
tail -3 gpurun_out/r03k_twin.log
