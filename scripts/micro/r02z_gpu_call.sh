#!/bin/bash
bash scripts/profile_round.sh twin 20 > gpurun_out/r02z_profile.log 2>&1
python scripts/pmc_traffic.py gpurun_out/prof_twin > gpurun_out/r02z_pmc_traffic.json 2>gpurun_out/r02z_pmc.err
tail -30 gpurun_out/r02z_profile.log | cut -c1-200; cat gpurun_out/r02z_pmc_traffic.json gpurun_out/r02z_pmc.err
rm -rf gpurun_out/prof_twin/stats gpurun_out/prof_twin/pmc_fetch gpurun_out/prof_twin/pmc_write
