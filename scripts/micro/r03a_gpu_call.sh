#!/bin/bash
mkdir -p gpurun_out
for tp in none 2016 2012 2008; do
  (timeout 600 python bench.py --config 3 --steps 20 --warmup 0 --no-cpu-baseline --no-saturated --twin-posteriors $tp 2>>gpurun_out/r03a.err | tail -1) > gpurun_out/r03a_cfg3_$tp.json
done
python - <<'PY'
import json
for tp in ['none','2016','2012','2008']:
    try:
        d=json.loads([l for l in open(f'gpurun_out/r03a_cfg3_{tp}.json') if l.startswith('{')][0])
        pp=d['config']['posteriors']
        print(tp, d['value'], d['seconds'], d['ess_per_sec'], {k:(v['clusters_per_chain'], round(v['ess_bulk_min'])) for k,v in pp.items()})
    except Exception as e: print(tp,'ERR',e)
PY
tail -3 gpurun_out/r03a.err
