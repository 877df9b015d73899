#!/bin/bash
mkdir -p gpurun_out
(timeout 600 python -m pytest tests/test_gpu_parity.py -q --tb=short -p no:cacheprovider -k "csv" 2>&1 | tail -4) > gpurun_out/r03m_csv.log
timeout 600 python scripts/micro/csv_probe.py 200 >> gpurun_out/r03m_csv.log 2>&1
nproc >> gpurun_out/r03m_csv.log
cat gpurun_out/r03m_csv.log
