#!/bin/bash
mkdir -p gpurun_out
timeout 900 python scripts/micro/dense_probe.py active 3:0,1,2:2,7,11:13,14,15:0,5,10:1,2,3 0,1,5 > gpurun_out/r02u_active.log 2>&1
cat gpurun_out/r02u_active.log
