#!/bin/bash
# block sums in LDS for launches of few samples per pixel (work blocks of 64 / 128 / 192 pixel-samples): whole suite + fuzz
python -m pytest tests -x -q -m gpu > gpurun_out/r05_pytest_gpu.txt 2>&1 || { tail -40 gpurun_out/r05_pytest_gpu.txt; exit 1; }
tail -1 gpurun_out/r05_pytest_gpu.txt
RTIOW_LARGE_BLOCK_MIN_ITEMS=0 FUZZ_LARGE=0.3 FUZZ_U53=0.2 FUZZ_HIGH_SPP=0.15 python tools/fuzz_parity.py 5000 99981 > gpurun_out/r05_fuzz_small.txt 2>&1; tail -1 gpurun_out/r05_fuzz_small.txt
RTIOW_SCAN_MODE=1 python tools/fuzz_parity.py 1000 99991 >> gpurun_out/r05_fuzz_small.txt 2>&1; tail -1 gpurun_out/r05_fuzz_small.txt
