#!/bin/bash
# diagnostic builds of the FINAL library: block counts, phase shares, exit times
( RTIOW_LIB=$PWD/tools/lib_counts.so python tools/block_counts.py; SCENE=cfg4 RTIOW_LIB=$PWD/tools/lib_counts.so python tools/block_counts.py
  echo "--- RT_COUNT_ENUM (counter 1 = enumeration trips, counter 6 = candidates pushed)"; RTIOW_LIB=$PWD/tools/var_cnt_enum.so python tools/block_counts.py; SCENE=cfg4 RTIOW_LIB=$PWD/tools/var_cnt_enum.so python tools/block_counts.py
  echo "--- RT_COUNT_ROWS (10k scene: counter 1 = footprint-row trips, counter 6 = list-emission trips)"; SCENE=cfg4 RTIOW_LIB=$PWD/tools/var_cnt_rows.so python tools/block_counts.py
  echo "--- RT_COUNT_REDRAW (counter 1 'camera block' = lanes that draw a unit-sphere sample, counter 6 'unit-sphere tries' = lanes that fail try 0; per pass)"; RTIOW_LIB=$PWD/tools/var_cnt_redraw.so python tools/block_counts.py; SCENE=cfg4 RTIOW_LIB=$PWD/tools/var_cnt_redraw.so python tools/block_counts.py ) > gpurun_out/r05_block_counts.txt 2>&1
( MODES=5 RTIOW_LIB=$PWD/tools/lib_stamps.so python tools/phase_shares.py; SCENE=cfg4 MODES=5 RTIOW_LIB=$PWD/tools/lib_stamps.so python tools/phase_shares.py ) > gpurun_out/r05_phase_shares.txt 2>&1
( RTIOW_LIB=$PWD/tools/lib_exit.so python tools/exit_times.py; SPP=500 RTIOW_LIB=$PWD/tools/lib_exit.so python tools/exit_times.py ) > gpurun_out/r05_exit_times.txt 2>&1
echo diag done
