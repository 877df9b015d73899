#!/bin/bash
mkdir -p gpurun_out

timeout 600 python scripts/micro/dense_probe.py pieces 2>&1 | grep "matrix pass" > gpurun_out/r02p_pieces.log
timeout 600 python scripts/micro/dense_probe.py sampler 8 30 > gpurun_out/r02p_sampler.log 2>&1
(timeout 900 python bench.py --config 4 --steps 4 --warmup 0 --no-cpu-baseline 2>gpurun_out/r02p_bench4.err | tail -1) > gpurun_out/r02p_bench_config4.json
tail -12 gpurun_out/r02p_dense.log; cat gpurun_out/r02p_pieces.log gpurun_out/r02p_sampler.log; cut -c1-200 gpurun_out/r02p_bench_config4.json
timeout 600 python scripts/micro/dense_probe.py sampler 3 30 >> gpurun_out/r02p_sampler.log 2>&1; tail -1 gpurun_out/r02p_sampler.log
