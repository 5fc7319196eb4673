#!/bin/bash
# Builds rtiow_amd/librtiow_hip.so for gfx950 (used by __graft_entry__.build()).
set -e
cd "$(dirname "$0")"
mkdir -p build
hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt \
  -fPIC -shared -Iinclude -Irtiow_amd/csrc "$@" \
  -o rtiow_amd/librtiow_hip.so rtiow_amd/csrc/rt_api.hip
