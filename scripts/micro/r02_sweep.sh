#!/bin/bash
# Round-2 probe (GPU box): us per leapfrog per chain of the 2016 posterior, 8 chains on clusters of 16, for partition
# weights (days weigh POTUS_CW_DAY, polls POTUS_CW_POLL when the days are dealt to the members) and with / without the
# late verdict (POTUS_NO_LATE_VERDICT=1: the previous leaf's verdicts taken in the poll phase, as in round 1).
for cw in "30 30" "30 45" "20 40" "10 50" "30 60" "40 20"; do
  set -- $cw
  echo "== K=16 cw_day=$1 cw_poll=$2"
  POTUS_CW_DAY=$1 POTUS_CW_POLL=$2 POTUS_K=16 timeout 200 python scripts/gpu_probe.py 240 8 2>&1 | head -1
done
echo "== K=16 30 30, verdicts in the poll phase"; POTUS_NO_LATE_VERDICT=1 POTUS_K=16 timeout 200 python scripts/gpu_probe.py 240 8 2>&1 | head -1
echo "== 16 chains K=16"; POTUS_K=16 timeout 200 python scripts/gpu_probe.py 240 16 2>&1 | head -1
if [ -f us_potus_model_amd/libpotus_hmc_prof.so ]; then
  echo "== in-kernel cycles, K=16"
  POTUS_LIB=$PWD/us_potus_model_amd/libpotus_hmc_prof.so POTUS_K=16 timeout 300 python scripts/gpu_probe.py 240 8 2>&1 | cut -c1-260
fi
