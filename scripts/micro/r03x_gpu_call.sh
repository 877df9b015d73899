#!/bin/bash
mkdir -p gpurun_out
{
echo "== tiles of 512 columns (as committed)"; timeout 600 python scripts/micro/dense_probe.py pieces 2>&1 | grep "matrix pass"
echo "== tiles of 1024 columns"; POTUS_LIB=$PWD/us_potus_model_amd/libpotus_hmc_ct1024.so timeout 600 python scripts/micro/dense_probe.py pieces 2>&1 | grep "matrix pass"
echo "== 1024: active sweep"; POTUS_LIB=$PWD/us_potus_model_amd/libpotus_hmc_ct1024.so timeout 600 python scripts/micro/dense_probe.py active 1:3:2,7,11:8:16 0 2>&1
} > gpurun_out/r03x_ct.log 2>&1
(POTUS_LIB=$PWD/us_potus_model_amd/libpotus_hmc_ct1024.so timeout 600 python -m pytest tests/test_gpu_dense.py -q --tb=short -p no:cacheprovider -k "product or pieces or matvec or launch_shape" 2>&1 | tail -3) >> gpurun_out/r03x_ct.log
cat gpurun_out/r03x_ct.log
