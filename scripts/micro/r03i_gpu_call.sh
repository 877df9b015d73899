#!/bin/bash
mkdir -p gpurun_out
(timeout 900 python -m pytest tests/test_gpu_twin.py -q --tb=short -p no:cacheprovider --timeout 400 -s 2>&1 | tail -40) > gpurun_out/r03i_twin.log
cat gpurun_out/r03i_twin.log | cut -c1-400
