#!/bin/bash
mkdir -p gpurun_out
(timeout 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider --deselect tests/test_gpu_boundary.py::test_watchdog_turns_a_missing_member_into_an_error 2>&1 | tail -80) > gpurun_out/r02a_pytest.log
(timeout 400 python bench.py 2>gpurun_out/r02a_bench.err | tail -1) > gpurun_out/r02a_bench.json
timeout 900 bash scripts/micro/r02_sweep.sh > gpurun_out/r02a_sweep.log 2>&1
(timeout 150 python -m pytest tests/test_gpu_boundary.py -q --tb=short -p no:cacheprovider -k watchdog 2>&1 | tail -25) > gpurun_out/r02a_watchdog.log
tail -5 gpurun_out/r02a_pytest.log; cut -c1-600 gpurun_out/r02a_bench.json; tail -3 gpurun_out/r02a_watchdog.log
