#!/bin/bash
mkdir -p gpurun_out
timeout 600 python scripts/micro/dense_debug.py > gpurun_out/r02g_debug.log 2>&1
cat gpurun_out/r02g_debug.log
