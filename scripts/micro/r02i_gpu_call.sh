#!/bin/bash
mkdir -p gpurun_out
(timeout 900 python -m pytest tests/test_gpu_dense.py -q --tb=short -p no:cacheprovider -s 2>&1 | tail -30) > gpurun_out/r02i_dense.log
timeout 600 python scripts/micro/dense_probe.py pieces > gpurun_out/r02i_dense_pieces.log 2>&1
tail -8 gpurun_out/r02i_dense.log; cat gpurun_out/r02i_dense_pieces.log
