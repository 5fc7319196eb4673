#!/bin/bash
# Two PMC passes on the render kernel of `bench.py <args>` (default: configs[1], 1200x675x100):
# instruction counts, then issue/wait cycles and memory-side write bytes.
#   tools/pmc_insts.sh [bench.py args...]
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/pmci
rm -rf $OUT; mkdir -p $OUT
ARGS="${@:---width 1200 --height 675 --spp 100}"
B="bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-other-configs $ARGS"
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_MFMA SQ_INSTS_BRANCH SQ_WAVE_CYCLES --output-format csv -d $OUT/p1 -- python3 $B > $OUT/p1.log 2>&1
echo "p1 rc=$?"
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/p2 -- python3 $B > $OUT/p2.log 2>&1
echo "p2 rc=$?"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/p3 -- python3 $B > $OUT/p3.log 2>&1
echo "p3 rc=$?"
python3 - <<'PY'
import csv, glob, collections
agg = collections.defaultdict(list)
for f in glob.glob("gpurun_out/pmci/p*/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "render_kernel" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
m = {k: sum(v) / len(v) for k, v in agg.items()}
print("  ".join(f"{k}={v/1e9:.4f}G" for k, v in sorted(m.items())))
if "SQ_ACTIVE_INST_VALU" in m and "GRBM_GUI_ACTIVE" in m:
    print("valu_busy = %.3f" % (m["SQ_ACTIVE_INST_VALU"] * 4 / (m["GRBM_GUI_ACTIVE"] / 8 * 1024)))
if "WRITE_SIZE" in m:
    print("WRITE_SIZE = %.1f MB per launch" % (m["WRITE_SIZE"] * 1024 / 1e6))
PY
