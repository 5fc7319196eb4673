#!/bin/bash
# final library of round 5: whole GPU suite, then the profiles (tools/profile_bench.sh); bench lines in a second call, after tools/summarize_profile.py
python -m pytest tests -x -q -m gpu > gpurun_out/r05_pytest_gpu.txt 2>&1 || { tail -40 gpurun_out/r05_pytest_gpu.txt; exit 1; }
tail -1 gpurun_out/r05_pytest_gpu.txt
bash tools/profile_bench.sh
