#!/bin/bash
# Two quick PMC passes of bench.py (instruction mix + wave-state counters) -> gpurun_out/pmcq/
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/pmcq
rm -rf $OUT; mkdir -p $OUT
ARGS="bench.py --steps 3 --warmup 1 --no-cpu-baseline"
i=0
for pmc in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS" \
           "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_BRANCH SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS"; do
  i=$((i+1))
  rocprofv3 --pmc $pmc --output-format csv -d $OUT/pmc$i -- python3 $ARGS > $OUT/pmc$i.log 2>&1
  echo "pmc$i rc=$?"
done
python3 - <<'PY'
import csv, glob, collections
pm = {}
for f in sorted(glob.glob("gpurun_out/pmcq/pmc*/*/*_counter_collection.csv")):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "render_kernel" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in agg.items(): pm[k] = sum(v) / len(v)
for k in sorted(pm): print(f"{k:28s} {pm[k]:18.0f}")
PY
