#!/bin/bash
mkdir -p gpurun_out
: > gpurun_out/r04b.log
for c in 12 16 4; do
  (timeout 300 python bench.py --steps 20 --warmup 0 --chains-per-gpu $c --no-cpu-baseline --no-saturated 2>>gpurun_out/r04b.err | tail -1) > gpurun_out/r04b_line.json
  python -c "
import json
d=json.loads([l for l in open('gpurun_out/r04b_line.json') if l.startswith('{')][0]); p=d['config']['posteriors']['2016']
print($c, 'chains, 1000 + 1000:', round(d['value']), 'leapfrogs/s,', round(d['seconds'],2), 's,', round(d['us_per_leapfrog_per_chain'],2), 'us per leapfrog per chain, ESS/s', round(d['ess_per_sec']), ', K', p['cus_per_chain'], 'x', p['clusters_per_chain'], 'rhat', round(d['rhat_max'],4))" >> gpurun_out/r04b.log
done
cat gpurun_out/r04b.log
