#!/bin/bash
mkdir -p gpurun_out
timeout 900 python scripts/micro/dense_probe.py active > gpurun_out/r02q_active.log 2>&1
cat gpurun_out/r02q_active.log
