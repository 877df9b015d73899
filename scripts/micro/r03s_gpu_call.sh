#!/bin/bash
mkdir -p gpurun_out
{
echo "== 12 chains, K=16 one cluster"; POTUS_K=16 timeout 200 python scripts/gpu_probe.py 240 12 2>&1 | head -1
echo "== 12 chains, K=10 twin, DW=8"; POTUS_K=10 POTUS_TWIN=1 timeout 200 python scripts/gpu_probe.py 240 12 2>&1 | head -1
echo "== 12 chains, K=10 twin, DW=4 (avg 26 days)"; POTUS_CL_DW4_MAXAVG=26 POTUS_K=10 POTUS_TWIN=1 timeout 200 python scripts/gpu_probe.py 240 12 2>&1 | head -1
echo "== 8 chains, K=16 twin (reference)"; POTUS_K=16 POTUS_TWIN=1 timeout 200 python scripts/gpu_probe.py 240 8 2>&1 | head -1
echo "== 16 chains, K=8 twin, DW=8"; POTUS_K=8 POTUS_TWIN=1 timeout 200 python scripts/gpu_probe.py 240 16 2>&1 | head -1
echo "== 16 chains, K=8 twin, DW=4 (32 days)"; POTUS_CL_DW4_MAXAVG=32 POTUS_K=8 POTUS_TWIN=1 timeout 200 python scripts/gpu_probe.py 240 16 2>&1 | head -1
echo "== 16 chains, K=16 one cluster"; POTUS_K=16 timeout 200 python scripts/gpu_probe.py 240 16 2>&1 | head -1
} > gpurun_out/r03s_k.log 2>&1
cat gpurun_out/r03s_k.log
