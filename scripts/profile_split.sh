#!/bin/bash
# PMC passes over the fp16x3 GEMM micro-benchmark (run on the GPU box from the repo root).
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/prof_split
mkdir -p $OUT
export ONLY="${ONLY:-s3_pw1 51200x2048x512 gelu}" CFGS="${CFGS:-12,0,10}" REPS=3
CMD="python scripts/split_bench.py"
rocprofv3 -L > $OUT/counters_avail.txt 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --kernel-trace -d $OUT/pmc_a -o sp -- $CMD > $OUT/a.log 2> $OUT/a.err
rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA --kernel-trace -d $OUT/pmc_b -o sp -- $CMD > $OUT/b.log 2> $OUT/b.err
for d in pmc_a pmc_b; do
  db=$(ls $OUT/$d/*/*.db 2>/dev/null | head -1)
  [ -n "$db" ] && python scripts/rocpd_summary.py $db --pmc $db > $OUT/$d.txt 2>&1
done
tail -5 $OUT/a.err $OUT/b.err
cat $OUT/pmc_a.txt $OUT/pmc_b.txt | grep -v "^$" | head -120
