#!/bin/bash
mkdir -p gpurun_out
(timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 2>gpurun_out/r02m_bench.err | tail -1) > gpurun_out/r02m_bench_line.json
hipcc --offload-arch=gfx950 -O3 -Wno-unused-value scripts/micro/mfma_vs_sparse.hip -o /tmp/mfma_vs_sparse > gpurun_out/r02m_mfma.log 2>&1 && /tmp/mfma_vs_sparse >> gpurun_out/r02m_mfma.log 2>&1
for lib in libpotus_hmc.so libpotus_hmc_rb256.so; do
  echo "== $lib" >> gpurun_out/r02m_rb.log
  POTUS_LIB=$PWD/us_potus_model_amd/$lib timeout 600 python scripts/micro/dense_probe.py pieces 2>&1 | grep "matrix pass" >> gpurun_out/r02m_rb.log
  POTUS_LIB=$PWD/us_potus_model_amd/$lib timeout 600 python scripts/micro/dense_probe.py sampler 8 30 >> gpurun_out/r02m_rb.log 2>&1
done
cut -c1-300 gpurun_out/r02m_bench_line.json; cat gpurun_out/r02m_mfma.log; cat gpurun_out/r02m_rb.log
