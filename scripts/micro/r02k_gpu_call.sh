#!/bin/bash
mkdir -p gpurun_out
timeout 1200 bash scripts/profile_dense.sh 8 30 > gpurun_out/r02k_profile_dense.log 2>&1
(timeout 900 python bench.py --config 4 --steps 4 --warmup 0 --no-cpu-baseline 2>gpurun_out/r02k_bench4.err | tail -1) > gpurun_out/r02k_bench_config4.json
(timeout 900 python bench.py --config 4 --steps 4 --warmup 0 --no-cpu-baseline --chains-per-gpu 8 2>>gpurun_out/r02k_bench4.err | tail -1) > gpurun_out/r02k_bench_config4_8chains.json
tail -40 gpurun_out/r02k_profile_dense.log | cut -c1-200; cut -c1-600 gpurun_out/r02k_bench_config4.json
