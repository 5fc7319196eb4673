#!/bin/bash
# One PMC pass (instruction counts by class) on the render kernel of configs[1] for each library given:
#   tools/pmc_quick.sh lib1.so lib2.so ...     (RTIOW_HIP_LIB selects the build the bench loads)
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
for LIB in "$@"; do
  OUT=gpurun_out/pmcq_$(basename $LIB .so)
  rm -rf $OUT; mkdir -p $OUT
  RTIOW_HIP_LIB=$PWD/$LIB rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_BRANCH SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM --output-format csv -d $OUT -- \
      python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-other-configs --width 1200 --height 675 --spp 100 > $OUT/run.log 2>&1
  python3 - "$OUT" "$LIB" <<'PY'
import csv, glob, collections, sys
agg = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "render_kernel" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
wb = 213985691 / 64.0          # wave-bounces of configs[1]
print(sys.argv[2], "  ".join(f"{k[9:]}={sum(v)/len(v)/wb:.1f}" for k, v in sorted(agg.items())), "per wave-bounce")
PY
done
