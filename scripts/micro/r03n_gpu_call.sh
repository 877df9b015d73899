#!/bin/bash
mkdir -p gpurun_out
: > gpurun_out/r03n_soak.log
for args in "--seed 1" "--seed 2 --chains-per-gpu 5" "--seed 3 --chains-per-gpu 7" "--seed 4 --chains-per-gpu 3" "--seed 5 --chains-per-gpu 6 --chunk 37"; do
  (timeout 600 python bench.py --steps 20 --warmup 0 --no-cpu-baseline --no-saturated $args 2>>gpurun_out/r03n.err | tail -1) > gpurun_out/r03n_line.json
  python - "$args" >> gpurun_out/r03n_soak.log <<'PY'
import json, sys
try:
    d=json.loads([l for l in open('gpurun_out/r03n_line.json') if l.startswith('{')][0])
    p=d['config']['posteriors']['2016']
    print(sys.argv[1], '->', round(d['value']), 'lf/s', 'rhat', round(d['rhat_max'],4), 'ess/s', round(d['ess_per_sec']), 'div', p['divergent_transitions'], 'status', p['chain_status'], 'clusters', p['clusters_per_chain'])
except Exception as e:
    print(sys.argv[1], 'FAILED', e)
PY
done
cat gpurun_out/r03n_soak.log; tail -3 gpurun_out/r03n.err
