#!/bin/bash
mkdir -p gpurun_out
(timeout 900 python bench.py --config 4 --steps 4 --warmup 0 --no-cpu-baseline 2>gpurun_out/r02r_bench4.err | tail -1) > gpurun_out/r02r_bench_config4.json
(timeout 900 python bench.py --config 4 --steps 4 --warmup 0 --no-cpu-baseline 2>>gpurun_out/r02r_bench4.err | tail -1) > gpurun_out/r02r_bench_config4_b.json
python - <<'PY'
import json
for f in ['gpurun_out/r02r_bench_config4.json','gpurun_out/r02r_bench_config4_b.json']:
    d=json.loads([l for l in open(f) if l.startswith('{')][0]); r=d['roofline']
    print(d['value'], d['leapfrogs'], d['seconds'], r['achieved'], r['avg_pass_ms'], r['leaf_rounds'], r['matrix_passes'])
PY
