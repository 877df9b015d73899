#!/bin/bash
mkdir -p gpurun_out
timeout 600 python scripts/micro/dense_probe.py pieces 2>&1 | grep "matrix pass" > gpurun_out/r02o_pieces.log
timeout 600 python scripts/micro/dense_probe.py sampler 8 30 > gpurun_out/r02o_sampler.log 2>&1
timeout 600 python scripts/micro/dense_probe.py sampler 2 30 >> gpurun_out/r02o_sampler.log 2>&1
cat gpurun_out/r02o_pieces.log gpurun_out/r02o_sampler.log
