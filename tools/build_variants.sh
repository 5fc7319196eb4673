#!/bin/bash
# Builds experimental variants of the library: tools/build_variants.sh NAME "-DFLAG ..." [NAME "-DFLAG" ...] -> tools/var_NAME.so
cd "$(dirname "$0")/.."
FLAGS="-O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -mllvm -amdgpu-mfma-vgpr-form -fPIC -shared -I include -I rtiow_amd/csrc"
while [ $# -ge 2 ]; do
  hipcc $FLAGS $2 -o tools/var_$1.so rtiow_amd/csrc/rt_api.hip &
  shift 2
done
wait
ls -la tools/var_*.so
