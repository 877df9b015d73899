#!/bin/bash
mkdir -p gpurun_out
(timeout 900 python -m pytest tests/test_gpu_twin.py -q --tb=short -p no:cacheprovider --timeout 400 -k "nine_to_eleven or default" 2>&1 | tail -6) > gpurun_out/r03r_twin.log
for c in 9 10 11 12; do
  (timeout 600 python bench.py --steps 6 --warmup 0 --chunk 50 --chains-per-gpu $c --no-cpu-baseline --no-saturated 2>>gpurun_out/r03r.err | tail -1) > gpurun_out/r03r_line.json
  python -c "
import json
d=json.loads([l for l in open('gpurun_out/r03r_line.json') if l.startswith('{')][0]); p=d['config']['posteriors']['2016']
print($c, 'chains:', round(d['value']), 'lf/s', round(d['us_per_leapfrog_per_chain'],2), 'us/leapfrog/chain, K', p['cus_per_chain'], 'clusters', p['clusters_per_chain'], 'status', set(p['chain_status']))" >> gpurun_out/r03r_twin.log
done
cat gpurun_out/r03r_twin.log
