#!/bin/bash
bash scripts/profile_dense.sh 8 30 > gpurun_out/r03w_profile_dense.log 2>&1
tail -40 gpurun_out/r03w_profile_dense.log | cut -c1-220
rm -rf gpurun_out/prof_dense/stats gpurun_out/prof_dense/pmc_fetch
