#!/bin/bash
mkdir -p gpurun_out
timeout 900 python scripts/micro/dense_probe.py active 2,7,11:0,5,10:1,4,6,9,12,15:3 0,1 > gpurun_out/r02v_active.log 2>&1
(timeout 900 python bench.py --config 4 --steps 4 --warmup 0 --no-cpu-baseline 2>gpurun_out/r02v_bench4.err | tail -1) > gpurun_out/r02v_bench_config4.json
timeout 600 python scripts/micro/dense_probe.py sampler 8 30 > gpurun_out/r02v_sampler.log 2>&1
(timeout 900 python -m pytest tests/test_gpu_dense.py -q --tb=short -p no:cacheprovider 2>&1 | tail -8) > gpurun_out/r02v_dense.log
cat gpurun_out/r02v_active.log gpurun_out/r02v_sampler.log; cut -c1-200 gpurun_out/r02v_bench_config4.json; tail -5 gpurun_out/r02v_dense.log
