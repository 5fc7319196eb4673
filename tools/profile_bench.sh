#!/bin/bash
# Profiles `bench.py` on the GPU box: kernel trace + stats, then PMC passes (each in its own run, no trace
# domains mixed with --pmc).  Three configurations: the bench default (1200x675x500, tag "target"),
# BASELINE configs[1] (1200x675x100, tag "cfg2"), configs[3] (10k spheres, 1920x1080x256, tag "tenk": the large-grid kernel) and
# configs[2] (3840x2160x500, tag "weak": the N = 1 half of the weak-scaling pair, `bench.py --weak-baseline`).  Output under gpurun_out/prof_<tag>/; summaries are copied
# into profiles/ by tools/summarize_profile.py.
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
run_cfg() {
  TAG=$1; shift
  OUT=gpurun_out/prof_$TAG
  rm -rf $OUT; mkdir -p $OUT
  STEPS=5; [ "$TAG" = weak ] && STEPS=2
  ARGS="bench.py --steps $STEPS --warmup 1 --no-cpu-baseline --no-other-configs --no-end-to-end $@"
  echo "$ARGS" > $OUT/command.txt
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ARGS > $OUT/trace.log 2>&1
  echo "$TAG trace rc=$?"
  i=0
  for pmc in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM" \
             "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_BF16 SQ_INSTS_BRANCH SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
             "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS" \
             "GRBM_GUI_ACTIVE GRBM_COUNT" \
             "FETCH_SIZE" "WRITE_SIZE" \
             "SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32"; do
    i=$((i+1))
    rocprofv3 --pmc $pmc --output-format csv -d $OUT/pmc$i -- python3 $ARGS > $OUT/pmc$i.log 2>&1
    echo "$TAG pmc$i rc=$? ($pmc)"
  done
}
run_cfg target
run_cfg cfg2 --width 1200 --height 675 --spp 100
run_cfg tenk --tenk
run_cfg weak --weak-baseline
