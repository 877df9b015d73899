#!/bin/bash
mkdir -p gpurun_out
{
for rep in 1 2; do
  echo "== two lanes per poll (rep $rep)"; POTUS_K=16 POTUS_TWIN=1 timeout 200 python scripts/gpu_probe.py 240 8 2>&1 | head -1
  echo "== one lane per poll (rep $rep)";  POTUS_LIB=$PWD/us_potus_model_amd/libpotus_hmc_pl1.so POTUS_K=16 POTUS_TWIN=1 timeout 200 python scripts/gpu_probe.py 240 8 2>&1 | head -1
done
echo "== one cluster: two lanes"; POTUS_K=16 timeout 200 python scripts/gpu_probe.py 240 8 2>&1 | head -1
echo "== one cluster: one lane";  POTUS_LIB=$PWD/us_potus_model_amd/libpotus_hmc_pl1.so POTUS_K=16 timeout 200 python scripts/gpu_probe.py 240 8 2>&1 | head -1
} > gpurun_out/r03f_pl.log 2>&1
(timeout 900 python -m pytest tests/test_gpu_parity.py -q --tb=short -p no:cacheprovider --timeout 400 -k "log_prob or oracle_chain or edge or stress" 2>&1 | tail -6) > gpurun_out/r03f_pytest.log
cat gpurun_out/r03f_pl.log; tail -4 gpurun_out/r03f_pytest.log
