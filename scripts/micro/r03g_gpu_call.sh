#!/bin/bash
mkdir -p gpurun_out
{
for rep in 1 2; do
  for v in 1 0 3 8; do
    lib=$PWD/us_potus_model_amd/libpotus_hmc_sl$v.so; [ $v = 1 ] && lib=$PWD/us_potus_model_amd/libpotus_hmc.so
    echo "== twin, s_sleep($v) between looks (rep $rep)"; POTUS_LIB=$lib POTUS_K=16 POTUS_TWIN=1 timeout 200 python scripts/gpu_probe.py 240 8 2>&1 | head -1
  done
done
} > gpurun_out/r03g_sleep.log 2>&1
(timeout 900 python -m pytest tests/test_gpu_twin.py -q --tb=short -p no:cacheprovider --timeout 400 -s -k many_seeds 2>&1 | tail -6) > gpurun_out/r03g_pytest.log
cat gpurun_out/r03g_sleep.log; tail -5 gpurun_out/r03g_pytest.log
