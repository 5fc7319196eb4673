cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r05_fuzz_knobs.txt
echo "# fuzz under the diagnostic knobs (every one selects another way of computing the SAME frame), final library $(python -c 'import bench; print(bench.kernel_source_sha())'); 2000 scenes each, FUZZ_LARGE=0.3 FUZZ_U53=0.2 FUZZ_HIGH_SPP=0.1 RTIOW_LARGE_BLOCK_MIN_ITEMS=0:" > $OUT
k=0
for knob in "RTIOW_GRID_DIM=3" "RTIOW_GRID_DIM=9" "RTIOW_NO_GRID=1" "RTIOW_BLOCKS_PER_CU=1" "RTIOW_BLOCKS_PER_CU=2" "RTIOW_RING_MIN_SPP=1000" "RTIOW_RING_MIN_SPP=12" "RTIOW_GRID_DIM=20"; do
  k=$((k+1))
  echo "# $knob" >> $OUT
  env $knob RTIOW_LARGE_BLOCK_MIN_ITEMS=0 FUZZ_LARGE=0.3 FUZZ_U53=0.2 FUZZ_HIGH_SPP=0.1 python tools/fuzz_parity.py 2000 $((88000 + k)) >> $OUT 2>&1
  grep "cases, " $OUT | grep -v "\.\.\." | tail -1
done
