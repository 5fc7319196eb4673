#!/bin/bash
# The hardware's view for tools/lds_conflicts.py: SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE / SQ_INSTS_LDS / SQ_WAIT_INST_LDS / SQ_WAVE_CYCLES of the
# shipped kernel on the same launches.   usage: tools/lds_conflicts.sh W H SPP [extra bench.py flags]
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
W=$1; H=$2; SPP=$3; shift 3
OUT=gpurun_out/ldsc_${W}x${H}x${SPP}
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_WAVE_CYCLES SQ_WAVES --output-format csv -d $OUT -- \
    python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-other-configs --width $W --height $H --spp $SPP "$@" > $OUT/run.log 2>&1
python3 - "$OUT" <<'PY'
import csv, glob, collections, sys
agg = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "render_kernel" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
m = {k: sum(v) / len(v) for k, v in agg.items()}
print(sys.argv[1], {k: round(v) for k, v in m.items()})
print("  conflicts / LDS-active cycles = %.3f;  LDS issue stalls / wave cycles = %.4f" % (m["SQ_LDS_BANK_CONFLICT"] / m["SQ_LDS_IDX_ACTIVE"], m["SQ_WAIT_INST_LDS"] / m["SQ_WAVE_CYCLES"]))
PY
