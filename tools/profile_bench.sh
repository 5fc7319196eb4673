#!/bin/bash
# Profiles `bench.py` on the GPU box: kernel trace + stats, then PMC passes (each in its own
# run, no trace domains mixed with --pmc).  Output under gpurun_out/prof/; summaries are
# copied into profiles/ by tools/summarize_profile.py.
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/prof
mkdir -p $OUT
ARGS="bench.py --steps 5 --warmup 1 --no-cpu-baseline"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ARGS > $OUT/trace.log 2>&1
echo "trace rc=$?"
i=0
for pmc in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM" \
           "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_BF16 SQ_INSTS_BRANCH SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS" \
           "GRBM_GUI_ACTIVE GRBM_COUNT" \
           "FETCH_SIZE" "WRITE_SIZE" \
           "SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32"; do
  i=$((i+1))
  rocprofv3 --pmc $pmc --output-format csv -d $OUT/pmc$i -- python3 $ARGS > $OUT/pmc$i.log 2>&1
  echo "pmc$i rc=$? ($pmc)"
done
