#!/bin/bash
mkdir -p gpurun_out
{
for k in 16 12 14 10 13 11; do
  echo "== K=$k one cluster"; POTUS_K=$k timeout 200 python scripts/gpu_probe.py 240 8 2>&1 | head -1
  echo "== K=$k twin"; POTUS_K=$k POTUS_TWIN=1 timeout 200 python scripts/gpu_probe.py 240 8 2>&1 | head -1
done
} > gpurun_out/r03q_k.log 2>&1
cat gpurun_out/r03q_k.log
