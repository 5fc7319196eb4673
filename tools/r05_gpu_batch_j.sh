#!/bin/bash
# two-deep, 16-slot ring for every small-grid kernel; block sums in LDS from 17 spp on it
python -m pytest tests -x -q -m gpu > gpurun_out/r05_pytest_gpu.txt 2>&1 || { tail -30 gpurun_out/r05_pytest_gpu.txt; exit 1; }
tail -1 gpurun_out/r05_pytest_gpu.txt
out=gpurun_out/r05_wide_ring.txt
: > $out
for spp in 100 500 32 20; do
  echo "== 1200x675x$spp  base | wide ring" >> $out; SPP=$spp python tools/abn.py tools/var_base.so tools/var_wide.so >> $out 2>&1
done
echo "== 1200x675x32 wide ring build, RTIOW_RING_MIN_SPP=37 (direct adds) | default (block sums in LDS)" >> $out
RTIOW_RING_MIN_SPP=37 SPP=32 python tools/abn.py tools/var_wide.so >> $out 2>&1
echo "== 10k spheres x32 (large-grid kernel: unchanged)" >> $out; python tools/abn_tenk.py tools/var_base.so tools/var_wide.so >> $out 2>&1
RTIOW_LARGE_BLOCK_MIN_ITEMS=0 FUZZ_LARGE=0.3 FUZZ_U53=0.2 FUZZ_HIGH_SPP=0.3 python tools/fuzz_parity.py 3000 99971 > gpurun_out/r05_fuzz_wide.txt 2>&1; tail -1 gpurun_out/r05_fuzz_wide.txt
