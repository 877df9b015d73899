#!/bin/bash
mkdir -p gpurun_out
(timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -x 2>&1 | grep -v "^Iteration\|^$" | tail -30) > gpurun_out/r02e_pytest.log
timeout 900 bash scripts/micro/r02_sweep.sh > gpurun_out/r02e_sweep.log 2>&1
tail -5 gpurun_out/r02e_pytest.log; head -40 gpurun_out/r02e_sweep.log | cut -c1-200
