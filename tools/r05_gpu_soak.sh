cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r05_fuzz_soak.txt
echo "# long soak on the final library (kernel sources $(python -c 'import bench; print(bench.kernel_source_sha())')):" > $OUT
echo "# RTIOW_LARGE_BLOCK_MIN_ITEMS=0 FUZZ_LARGE=0.3 FUZZ_U53=0.3 FUZZ_HIGH_SPP=0.1 fuzz_parity.py 12000 99001:" >> $OUT
RTIOW_LARGE_BLOCK_MIN_ITEMS=0 FUZZ_LARGE=0.3 FUZZ_U53=0.3 FUZZ_HIGH_SPP=0.1 python tools/fuzz_parity.py 12000 99001 >> $OUT 2>&1
echo "# RTIOW_SCAN_MODE=1 fuzz_parity.py 1500 99501 (the in-order exact scan behind the VALU filter):" >> $OUT
RTIOW_SCAN_MODE=1 python tools/fuzz_parity.py 1500 99501 >> $OUT 2>&1
echo "# FUZZ_U53=1.0 FUZZ_LARGE=0.3 fuzz_parity.py 2500 99701 (every case with RT_FLAG_UNIFORM53):" >> $OUT
FUZZ_U53=1.0 FUZZ_LARGE=0.3 python tools/fuzz_parity.py 2500 99701 >> $OUT 2>&1
grep -v amdgpu $OUT | grep "cases, \|^#" | grep -v "\.\.\."
python tools/grid_dim_sweep.py 0 3 4 5 6 16 17 18 19 20 21 22 24 > gpurun_out/r05_grid_dim_sweep.txt 2>&1
cat gpurun_out/r05_grid_dim_sweep.txt
