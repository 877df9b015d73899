#!/bin/bash
mkdir -p gpurun_out
(timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 2>gpurun_out/r03l_bench.err | tail -1) > gpurun_out/r03l_bench.json
(timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2) > gpurun_out/r03l_smoke.log
(timeout 1800 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider --timeout 600 2>&1 | tail -12) > gpurun_out/r03l_pytest.log
cut -c1-200 gpurun_out/r03l_bench.json; cat gpurun_out/r03l_smoke.log; tail -4 gpurun_out/r03l_pytest.log
