#!/bin/bash
mkdir -p gpurun_out
(timeout 2400 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -s 2>&1 | grep -v "^Iteration\|^$" | tail -40) > gpurun_out/r02l_pytest.log
hipcc --offload-arch=gfx950 -O3 -Wno-unused-value scripts/micro/mfma_vs_sparse.hip -o /tmp/mfma_vs_sparse > gpurun_out/r02l_mfma.log 2>&1 && /tmp/mfma_vs_sparse >> gpurun_out/r02l_mfma.log 2>&1
timeout 600 python scripts/micro/dense_probe.py sampler 8 30 > gpurun_out/r02l_dense_sampler.log 2>&1
(timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 2>gpurun_out/r02l_bench.err | tail -1) > gpurun_out/r02l_bench_line.json
tail -12 gpurun_out/r02l_pytest.log; cat gpurun_out/r02l_mfma.log; cat gpurun_out/r02l_dense_sampler.log; cut -c1-300 gpurun_out/r02l_bench_line.json
