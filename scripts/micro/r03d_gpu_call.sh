#!/bin/bash
mkdir -p gpurun_out
(timeout 900 python -m pytest tests/test_gpu_dense.py -q --tb=short -p no:cacheprovider 2>&1 | tail -6) > gpurun_out/r03d_dense.log
timeout 600 python scripts/micro/dense_probe.py pieces 2>&1 | grep "matrix pass" > gpurun_out/r03d_pieces.log
timeout 900 python scripts/micro/dense_probe.py active 1:3:2,7,11:8:16 0 > gpurun_out/r03d_active.log 2>&1
timeout 600 python scripts/micro/dense_probe.py sampler 8 30 > gpurun_out/r03d_sampler.log 2>&1
(timeout 600 python bench.py --config 4 --steps 4 --warmup 0 --no-cpu-baseline 2>gpurun_out/r03d_bench.err | tail -1) > gpurun_out/r03d_bench4.json
tail -3 gpurun_out/r03d_dense.log; cat gpurun_out/r03d_pieces.log gpurun_out/r03d_active.log; cut -c1-420 gpurun_out/r03d_sampler.log; python -c "
import json
d=json.loads([l for l in open('gpurun_out/r03d_bench4.json') if l.startswith('{')][0]); print(d['value'], d['roofline']['frac'], d['roofline']['avg_pass_ms'])"
