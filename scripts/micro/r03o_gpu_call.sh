#!/bin/bash
mkdir -p gpurun_out
for rep in 1 2; do
(timeout 900 python bench.py --steps 20 --warmup 0 --no-cpu-baseline --no-saturated 2>gpurun_out/r03o_bench.err | tail -1) > gpurun_out/r03o_bench$rep.json
done
python -c "
import json
for r in (1,2):
    d=json.loads([l for l in open('gpurun_out/r03o_bench%d.json'%r) if l.startswith('{')][0]); print(d['value'], d['seconds'], d['ess_per_sec'], d['sampling_seconds'], d['config']['posteriors']['2016']['twin']['leaves_run_per_counted'])"
