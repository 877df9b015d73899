#!/bin/bash
mkdir -p gpurun_out
(timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 2>gpurun_out/r02d_bench.err | tail -1) > gpurun_out/r02d_bench_line.json
timeout 600 bash scripts/profile_round.sh cl 20 > gpurun_out/r02d_profile_cl.log 2>&1
timeout 600 python scripts/micro/dense_probe.py pieces > gpurun_out/r02d_dense_pieces.log 2>&1
timeout 900 bash scripts/profile_dense.sh 8 30 > gpurun_out/r02d_profile_dense.log 2>&1
timeout 300 python scripts/micro/stress_probe.py 60 16,8 > gpurun_out/r02d_stress_probe.log 2>&1
(timeout 600 python bench.py --config 3 --steps 20 --no-cpu-baseline --no-saturated 2>gpurun_out/r02d_bench3.err | tail -1) > gpurun_out/r02d_bench_config3.json
(timeout 900 python bench.py --config 4 --steps 4 --warmup 0 --no-cpu-baseline 2>gpurun_out/r02d_bench4.err | tail -1) > gpurun_out/r02d_bench_config4.json
cut -c1-300 gpurun_out/r02d_bench_line.json; cat gpurun_out/r02d_dense_pieces.log; cat gpurun_out/r02d_stress_probe.log; cut -c1-400 gpurun_out/r02d_bench_config4.json; tail -3 gpurun_out/r02d_bench4.err
