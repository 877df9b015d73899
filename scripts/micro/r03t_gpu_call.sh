#!/bin/bash
mkdir -p gpurun_out
(timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 2>gpurun_out/r03t_bench.err | tail -1) > gpurun_out/r03t_bench.json
(timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2) > gpurun_out/r03t_smoke.log
(timeout 1800 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider --timeout 600 2>&1 | tail -12) > gpurun_out/r03t_pytest.log
(timeout 600 python bench.py --config 3 --steps 20 --warmup 0 --no-cpu-baseline --no-saturated 2>>gpurun_out/r03t_bench.err | tail -1) > gpurun_out/r03t_bench3.json
cut -c1-200 gpurun_out/r03t_bench.json; cat gpurun_out/r03t_smoke.log; tail -4 gpurun_out/r03t_pytest.log
python -c "
import json
d=json.loads([l for l in open('gpurun_out/r03t_bench.json') if l.startswith('{')][0]); print(d['value'], d['seconds'], d['ess_per_sec'], d['roofline']['frac'], d['roofline']['traffic'], d['config']['posteriors']['2016']['twin']['leaves_run_per_counted'], d['cpu_baseline']['value'])
d=json.loads([l for l in open('gpurun_out/r03t_bench3.json') if l.startswith('{')][0]); print(d['value'], d['seconds'])"
