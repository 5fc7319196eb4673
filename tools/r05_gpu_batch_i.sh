#!/bin/bash
# final library of round 5 (small-grid kernel: ring of 2 x 16 pixel slots for the large work blocks, large blocks from 69 spp): whole GPU suite, fuzz, profiles
python -m pytest tests -x -q -m gpu > gpurun_out/r05_pytest_gpu.txt 2>&1 || { tail -30 gpurun_out/r05_pytest_gpu.txt; exit 1; }
tail -1 gpurun_out/r05_pytest_gpu.txt
RTIOW_LARGE_BLOCK_MIN_ITEMS=0 FUZZ_LARGE=0.3 FUZZ_U53=0.2 FUZZ_HIGH_SPP=0.4 python tools/fuzz_parity.py 3000 99961 > gpurun_out/r05_fuzz_ring2.txt 2>&1 || { tail -5 gpurun_out/r05_fuzz_ring2.txt; exit 1; }
tail -1 gpurun_out/r05_fuzz_ring2.txt
bash tools/profile_bench.sh
