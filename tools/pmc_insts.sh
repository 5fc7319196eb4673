#!/bin/bash
# One PMC pass: instruction counts of the render kernel (bench.py, 3 steps)
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/pmci
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_MFMA SQ_INSTS_BRANCH SQ_WAVE_CYCLES --output-format csv -d $OUT/p -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/p.log 2>&1
echo "rc=$?"
python3 - <<'PY'
import csv, glob, collections
agg = collections.defaultdict(list)
for f in glob.glob("gpurun_out/pmci/p/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "render_kernel" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
print("  ".join(f"{k}={sum(v)/len(v)/1e9:.3f}G" for k, v in sorted(agg.items())))
PY
