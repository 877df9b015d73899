#!/bin/bash
mkdir -p gpurun_out
(timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider --timeout 600 2>&1 | tail -40) > gpurun_out/r02y_pytest.log
cat gpurun_out/r02y_pytest.log
