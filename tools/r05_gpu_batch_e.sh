#!/bin/bash
# What does the work-block size change?  1200x675x500 with blocks of 256 (threshold never reached) and of 1 024 (threshold 0):
# block execution counts and stamped phase shares of both.
out=gpurun_out/r05_block_size_counts.txt
: > $out
for lb in 999999999999 0; do
  echo "=== RTIOW_LARGE_BLOCK_MIN_ITEMS=$lb (999999999999: blocks of 256; 0: blocks of 1024), 1200x675x500" >> $out
  RTIOW_LARGE_BLOCK_MIN_ITEMS=$lb SPP=500 RTIOW_LIB=$PWD/tools/lib_counts.so python tools/block_counts.py >> $out 2>&1
  RTIOW_LARGE_BLOCK_MIN_ITEMS=$lb SPP=500 MODES=5 RTIOW_LIB=$PWD/tools/lib_stamps.so python tools/phase_shares.py >> $out 2>&1
done
