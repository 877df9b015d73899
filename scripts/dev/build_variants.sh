#!/bin/bash
# Development: build several variants of libpotus_hmc.so side by side (build/variants/libpotus_<name>.so), in parallel.
#   scripts/dev/build_variants.sh name1:"-DA=1 -DB=0" name2:"-DPOTUS_PROF" ...
# The variants travel to the GPU box with the snapshot; scripts/gpu_probe.py picks one through POTUS_LIB.
cd "$(dirname "$0")/../.."
mkdir -p build/variants
pids=()
for spec in "$@"; do
  name="${spec%%:*}"; flags="${spec#*:}"
  ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wno-unused-result $flags \
      us_potus_model_amd/csrc/potus_hmc.hip -o build/variants/libpotus_$name.so > build/variants/$name.log 2>&1 \
      && echo "built $name" || { echo "FAILED $name"; tail -5 build/variants/$name.log; } ) &
  pids+=($!)
done
for p in "${pids[@]}"; do wait $p; done
