#!/bin/bash
mkdir -p gpurun_out
(timeout 600 python bench.py --config 2 --steps 4 --warmup 1 --no-saturated 2>gpurun_out/r03v.err | tail -1) > gpurun_out/r03v_cfg2.json
(MASTER_ADDR=127.0.0.1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --steps 4 --warmup 0 --no-cpu-baseline --no-saturated 2>>gpurun_out/r03v.err | tail -1) > gpurun_out/r03v_torchrun.json
python -c "
import json
for f in ['gpurun_out/r03v_cfg2.json','gpurun_out/r03v_torchrun.json']:
    try:
        d=json.loads([l for l in open(f) if l.startswith('{')][0]); print(f, d['value'], d['n_gpus'], d['config']['workload'][:90], 'cpu' in str(d.get('cpu_baseline',''))[:5], d.get('cpu_baseline',{}).get('value'))
    except Exception as e: print(f, 'ERR', e)"
tail -3 gpurun_out/r03v.err
