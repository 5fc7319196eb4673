cd $GRAFT_REPO_ROOT
( echo "--- RT_COUNT_REDRAW (counter 1 'camera block' = lanes that draw a unit-sphere sample, counter 6 'unit-sphere tries' = lanes that fail try 0; per pass)"; RTIOW_LIB=$PWD/tools/var_cnt_redraw.so python tools/block_counts.py; SCENE=cfg4 RTIOW_LIB=$PWD/tools/var_cnt_redraw.so python tools/block_counts.py ) > gpurun_out/r05_block_counts_redraw.txt 2>&1
grep "camera block\|unit-sphere tries" gpurun_out/r05_block_counts_redraw.txt
bash tools/profile_bench.sh > gpurun_out/r05_profile_bench.log 2>&1; grep -c "rc=0" gpurun_out/r05_profile_bench.log
