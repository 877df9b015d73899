#!/bin/bash
mkdir -p gpurun_out
timeout 900 python scripts/micro/dense_probe.py active 1,-1,2,-2,4,-4,8,-8 0,8 > gpurun_out/r02s_active.log 2>&1
cat gpurun_out/r02s_active.log
