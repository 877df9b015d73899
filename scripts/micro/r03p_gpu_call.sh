#!/bin/bash
mkdir -p gpurun_out
(timeout 900 python -m pytest tests/test_gpu_twin.py -q --tb=short -p no:cacheprovider --timeout 400 2>&1 | tail -4) > gpurun_out/r03p_twin.log
bash scripts/profile_round.sh twin 20 > gpurun_out/r03p_profile.log 2>&1
python scripts/pmc_traffic.py gpurun_out/prof_twin > gpurun_out/r03p_pmc_traffic.json 2>gpurun_out/r03p_pmc.err
tail -3 gpurun_out/r03p_twin.log; grep -A3 "stats/r_results" gpurun_out/prof_twin/summary.txt | cut -c1-160; cat gpurun_out/r03p_pmc_traffic.json gpurun_out/r03p_pmc.err
rm -rf gpurun_out/prof_twin/stats gpurun_out/prof_twin/pmc_fetch gpurun_out/prof_twin/pmc_write
