#!/bin/bash
mkdir -p gpurun_out
(timeout 1800 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider --timeout 600 2>&1 | tail -8) > gpurun_out/r04a_pytest.log
tail -4 gpurun_out/r04a_pytest.log
