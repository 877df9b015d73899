#!/bin/bash
mkdir -p gpurun_out
{
echo "== K=4, 64 chains"; POTUS_K=4 timeout 300 python scripts/gpu_probe.py 90 64 2>&1 | tail -3
echo "== K=1, 64 chains"; POTUS_K=1 timeout 300 python scripts/gpu_probe.py 90 64 2>&1 | head -1
echo "== K=8, 32 chains"; POTUS_K=8 timeout 300 python scripts/gpu_probe.py 90 32 2>&1 | head -1
echo "== K=4, 32 chains"; POTUS_K=4 timeout 300 python scripts/gpu_probe.py 90 32 2>&1 | head -1
} > gpurun_out/r03b_k4.log 2>&1
cat gpurun_out/r03b_k4.log
