#!/bin/bash
# Diagnostic builds of the library (never shipped): wave-time per phase (RT_PHASE_STAMPS), block
# execution counts (RT_BLOCK_COUNTS; with -DRT_COUNT_ROWS counters 1 and 6 count the large grid's footprint-row and list-emission trips instead of
# camera blocks and unit-sphere tries, with -DRT_COUNT_ENUM the enumeration's trips and the candidates it pushes: tools/build_variants.sh NAME "-DRT_BLOCK_COUNTS -DRT_COUNT_ENUM"),
# wave exit times (RT_EXIT_TIMES).  -> tools/lib_{stamps,counts,exit}.so
cd "$(dirname "$0")/.."
FLAGS="-O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -mllvm -amdgpu-mfma-vgpr-form -fPIC -shared -I include -I rtiow_amd/csrc"
hipcc $FLAGS -DRT_PHASE_STAMPS -o tools/lib_stamps.so rtiow_amd/csrc/rt_api.hip &
hipcc $FLAGS -DRT_BLOCK_COUNTS -o tools/lib_counts.so rtiow_amd/csrc/rt_api.hip &
hipcc $FLAGS -DRT_EXIT_TIMES -o tools/lib_exit.so rtiow_amd/csrc/rt_api.hip &
hipcc $FLAGS -DRT_LDS_CONFLICTS -o tools/lib_ldsc.so rtiow_amd/csrc/rt_api.hip &
wait
ls -la tools/lib_stamps.so tools/lib_counts.so tools/lib_exit.so
