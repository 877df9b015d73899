#!/bin/bash
mkdir -p gpurun_out
(timeout 900 python -m pytest tests/test_gpu_twin.py -x -q --tb=short -p no:cacheprovider --timeout 240 -s 2>&1 | tail -60) > gpurun_out/r02w_twin.log
cat gpurun_out/r02w_twin.log
