#!/bin/bash
# ring of 2 x 16 pixel slots for the large work blocks (large blocks from 69 spp on)
python -m pytest tests/test_gpu_parity.py tests/test_gpu_progressive.py -x -q -m gpu > gpurun_out/r05_ring2_pytest.txt 2>&1 || { tail -30 gpurun_out/r05_ring2_pytest.txt; exit 1; }
tail -1 gpurun_out/r05_ring2_pytest.txt
out=gpurun_out/r05_ring2.txt
: > $out
echo "== 1200x675x500 (large blocks in both)" >> $out; SPP=500 python tools/abn.py tools/var_base.so tools/var_ring2.so >> $out 2>&1
echo "== 1200x675x100 (blocks of 256 in both)" >> $out; python tools/abn.py tools/var_base.so tools/var_ring2.so >> $out 2>&1
echo "== 1200x675x100, RTIOW_LARGE_BLOCK_MIN_ITEMS=0 (ring2: blocks of 1024; base: spp < 147 keeps 256)" >> $out; RTIOW_LARGE_BLOCK_MIN_ITEMS=0 python tools/abn.py tools/var_base.so tools/var_ring2.so >> $out 2>&1
echo "== 10k spheres x32 (256 in both)" >> $out; python tools/abn_tenk.py tools/var_base.so tools/var_ring2.so >> $out 2>&1
echo "== progressive, default thresholds" >> $out; python tools/progressive_bench.py 5 >> $out 2>&1
echo "== progressive, RTIOW_LARGE_BLOCK_MIN_ITEMS=0 (the 100-spp and 250-spp passes on blocks of 1024)" >> $out; RTIOW_LARGE_BLOCK_MIN_ITEMS=0 python tools/progressive_bench.py 5 >> $out 2>&1
