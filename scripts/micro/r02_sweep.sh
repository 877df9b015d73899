#!/bin/bash
# Round-2 probe (GPU box): us per leapfrog per chain of the 2016 posterior, 8 chains, for cluster sizes and partition
# weights (days weigh POTUS_CW_DAY, polls POTUS_CW_POLL when the days are dealt to the members).
for K in 16 32; do
  for cw in "51 10" "51 20" "51 30" "51 45" "30 30"; do
    set -- $cw
    echo "== K=$K cw_day=$1 cw_poll=$2"
    POTUS_CW_DAY=$1 POTUS_CW_POLL=$2 POTUS_K=$K timeout 200 python scripts/gpu_probe.py 240 8 2>&1 | head -1
  done
done
echo "== 16 chains K=16"; POTUS_K=16 timeout 200 python scripts/gpu_probe.py 240 16 2>&1 | head -1
if [ -f us_potus_model_amd/libpotus_hmc_prof.so ]; then
  for K in 16 32; do
    echo "== in-kernel cycles, K=$K"
    POTUS_LIB=$PWD/us_potus_model_amd/libpotus_hmc_prof.so POTUS_K=$K timeout 300 python scripts/gpu_probe.py 240 8 2>&1 | cut -c1-260
  done
fi
