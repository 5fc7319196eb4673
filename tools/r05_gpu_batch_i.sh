#!/bin/bash
# final library of round 5: whole GPU suite, fuzz, then the profiles (tools/profile_bench.sh); bench lines in a second call, after tools/summarize_profile.py
python -m pytest tests -x -q -m gpu > gpurun_out/r05_pytest_gpu.txt 2>&1 || { tail -40 gpurun_out/r05_pytest_gpu.txt; exit 1; }
tail -1 gpurun_out/r05_pytest_gpu.txt
RTIOW_LARGE_BLOCK_MIN_ITEMS=0 FUZZ_LARGE=0.4 FUZZ_U53=0.2 FUZZ_HIGH_SPP=0.15 python tools/fuzz_parity.py 4000 99983 > gpurun_out/r05_fuzz_final.txt 2>&1 || { tail -5 gpurun_out/r05_fuzz_final.txt; exit 1; }
tail -1 gpurun_out/r05_fuzz_final.txt
bash tools/profile_bench.sh
