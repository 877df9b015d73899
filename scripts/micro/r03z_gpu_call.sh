#!/bin/bash
mkdir -p gpurun_out
{
for v in 2 0 1 3; do
  lib=$PWD/us_potus_model_amd/libpotus_hmc_aux$v.so; [ $v = 2 ] && lib=$PWD/us_potus_model_amd/libpotus_hmc.so
  echo "== matrix loads with cache policy aux=$v (0 default, 1 sc0, 2 nt, 3 sc0+nt)"; POTUS_LIB=$lib timeout 300 python scripts/micro/dense_probe.py pieces 2>&1 | grep "matrix pass" | grep -v "nrhs=1\|nrhs=3" | cut -c1-110
done
} > gpurun_out/r03z_aux.log 2>&1
cat gpurun_out/r03z_aux.log
