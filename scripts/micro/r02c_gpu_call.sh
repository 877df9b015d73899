#!/bin/bash
mkdir -p gpurun_out
(timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -s 2>&1 | grep -v "^Iteration\|^$" | tail -60) > gpurun_out/r02c_pytest.log
tail -25 gpurun_out/r02c_pytest.log
