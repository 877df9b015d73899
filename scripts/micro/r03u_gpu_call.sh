#!/bin/bash
mkdir -p gpurun_out
(timeout 900 python bench.py --steps 20 --warmup 0 --no-cpu-baseline --no-saturated 2>gpurun_out/r03u_bench.err | tail -1) > gpurun_out/r03u_bench.json
(timeout 900 python -m pytest tests/test_gpu_twin.py tests/test_gpu_boundary.py -q --tb=short -p no:cacheprovider --timeout 400 2>&1 | tail -4) > gpurun_out/r03u_twin.log
cut -c1-160 gpurun_out/r03u_bench.json; tail -3 gpurun_out/r03u_twin.log
