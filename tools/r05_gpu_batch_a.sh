cd $GRAFT_REPO_ROOT
python -m pytest tests -q -m gpu > gpurun_out/r05_pytest_gpu.txt 2>&1; tail -3 gpurun_out/r05_pytest_gpu.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu
( RTIOW_LIB=$PWD/tools/lib_counts.so python tools/block_counts.py; SCENE=cfg4 RTIOW_LIB=$PWD/tools/lib_counts.so python tools/block_counts.py; echo "--- RT_COUNT_ENUM (counter 1 = enumeration trips, counter 6 = candidates pushed)"; RTIOW_LIB=$PWD/tools/var_cnt_enum.so python tools/block_counts.py; SCENE=cfg4 RTIOW_LIB=$PWD/tools/var_cnt_enum.so python tools/block_counts.py; echo "--- RT_COUNT_ROWS (10k scene: counter 1 = footprint-row trips, counter 6 = list-emission trips)"; SCENE=cfg4 RTIOW_LIB=$PWD/tools/var_cnt_rows.so python tools/block_counts.py ) > gpurun_out/r05_block_counts.txt 2>&1
( MODES=5 python tools/phase_shares.py; SCENE=cfg4 MODES=5 python tools/phase_shares.py ) > gpurun_out/r05_phase_shares.txt 2>&1
( RTIOW_LIB=$PWD/tools/lib_exit.so python tools/exit_times.py; SPP=500 RTIOW_LIB=$PWD/tools/lib_exit.so python tools/exit_times.py ) > gpurun_out/r05_exit_times.txt 2>&1
echo diag done
python bench.py --steps 20 --warmup 5 > gpurun_out/r05_bench_n1.json 2> gpurun_out/r05_bench_n1.err; echo bench rc=$?
python bench.py --tenk --steps 10 --warmup 2 > gpurun_out/r05_bench_tenk.json 2> gpurun_out/r05_bench_tenk.err; echo tenk rc=$?
python bench.py --weak-baseline --steps 3 --warmup 1 > gpurun_out/r05_bench_weak_baseline.json 2> gpurun_out/r05_bench_weak.err; echo weak rc=$?
RTIOW_LARGE_BLOCK_MIN_ITEMS=0 FUZZ_LARGE=0.3 FUZZ_U53=0.2 FUZZ_HIGH_SPP=0.1 python tools/fuzz_parity.py 6000 97001 > gpurun_out/r05_fuzz_parity_raw.txt 2>&1; tail -2 gpurun_out/r05_fuzz_parity_raw.txt
