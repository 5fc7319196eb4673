cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r05_fuzz_soak.txt
echo "# long soak on the final library (kernel sources $(python -c 'import bench; print(bench.kernel_source_sha())')):" > $OUT
echo "# RTIOW_LARGE_BLOCK_MIN_ITEMS=0 FUZZ_LARGE=0.3 FUZZ_U53=0.3 FUZZ_HIGH_SPP=0.1 fuzz_parity.py 12000 99002 (40 % of the cases with 5..89 spp: block sums in LDS on work blocks of 64..256):" >> $OUT
RTIOW_LARGE_BLOCK_MIN_ITEMS=0 FUZZ_LARGE=0.3 FUZZ_U53=0.3 FUZZ_HIGH_SPP=0.1 python tools/fuzz_parity.py 12000 99002 >> $OUT 2>&1
echo "# RTIOW_SCAN_MODE=1 fuzz_parity.py 1500 99502 (the in-order exact scan behind the VALU filter):" >> $OUT
RTIOW_SCAN_MODE=1 python tools/fuzz_parity.py 1500 99502 >> $OUT 2>&1
echo "# FUZZ_U53=1.0 FUZZ_LARGE=0.3 fuzz_parity.py 2500 99702 (every case with RT_FLAG_UNIFORM53):" >> $OUT
FUZZ_U53=1.0 FUZZ_LARGE=0.3 python tools/fuzz_parity.py 2500 99702 >> $OUT 2>&1
grep -v amdgpu $OUT | grep "cases, \|^#" | grep -v "\.\.\."
