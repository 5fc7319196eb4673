#!/bin/bash
# Builds the library as it was at a git revision (kernel sources only) for A/B timing: tools/build_at.sh REV NAME -> tools/var_NAME.so
set -e
cd "$(dirname "$0")/.."
REV=$1; NAME=$2
TMP=$(mktemp -d)
mkdir -p $TMP/csrc $TMP/include
for f in rt_api.hip rt_device.hpp rt_kernels.hpp; do git show $REV:rtiow_amd/csrc/$f > $TMP/csrc/$f; done
git show $REV:include/rtiow_hip.h > $TMP/include/rtiow_hip.h
hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -mllvm -amdgpu-mfma-vgpr-form \
  -fPIC -shared -I $TMP/include -I $TMP/csrc -o tools/var_$NAME.so $TMP/csrc/rt_api.hip
rm -rf $TMP
ls -la tools/var_$NAME.so
