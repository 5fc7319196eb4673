cd $GRAFT_REPO_ROOT
python -m pytest tests -q -m gpu > gpurun_out/r05_pytest_gpu.txt 2>&1; tail -2 gpurun_out/r05_pytest_gpu.txt
python bench.py --steps 20 --warmup 5 > gpurun_out/r05_bench_n1.json 2> gpurun_out/r05_bench_n1.err; echo bench rc=$?
python bench.py --tenk --steps 10 --warmup 2 > gpurun_out/r05_bench_tenk.json 2> gpurun_out/r05_bench_tenk.err; echo tenk rc=$?
python bench.py --weak-baseline --steps 3 --warmup 1 > gpurun_out/r05_bench_weak_baseline.json 2> gpurun_out/r05_bench_weak.err; echo weak rc=$?
bash tools/profile_bench.sh > gpurun_out/r05_profile_bench.log 2>&1; grep -c "rc=0" gpurun_out/r05_profile_bench.log
RTIOW_LARGE_BLOCK_MIN_ITEMS=0 FUZZ_LARGE=0.3 FUZZ_U53=0.2 FUZZ_HIGH_SPP=0.1 python tools/fuzz_parity.py 2000 98001 > gpurun_out/r05_fuzz_parity_final.txt 2>&1; tail -1 gpurun_out/r05_fuzz_parity_final.txt
