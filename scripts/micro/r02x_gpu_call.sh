#!/bin/bash
mkdir -p gpurun_out
(timeout 900 python bench.py --steps 20 --warmup 0 2>gpurun_out/r02x_bench.err | tail -1) > gpurun_out/r02x_bench_twin.json
(timeout 900 python bench.py --steps 20 --warmup 0 --twin 0 --no-cpu-baseline --no-saturated 2>>gpurun_out/r02x_bench.err | tail -1) > gpurun_out/r02x_bench_one.json
python - <<'PY'
import json
for f in ['gpurun_out/r02x_bench_twin.json','gpurun_out/r02x_bench_one.json']:
    try:
        d=json.loads([l for l in open(f) if l.startswith('{')][0])
        print(f, d['value'], d['seconds'], d['ess_per_sec'], d['rhat_max'], d['us_per_leapfrog_per_chain'], d['config']['parallelism'][-90:])
    except Exception as e: print(f, 'ERR', e)
PY
tail -5 gpurun_out/r02x_bench.err
