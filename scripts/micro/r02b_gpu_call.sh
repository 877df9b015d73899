#!/bin/bash
mkdir -p gpurun_out
(timeout 900 python -m pytest tests/test_gpu_dense.py -q --tb=short -p no:cacheprovider -x -s 2>&1 | tail -60) > gpurun_out/r02b_dense.log
tail -30 gpurun_out/r02b_dense.log
