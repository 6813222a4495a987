#!/bin/bash
# PMC passes over the pre-split fp16x3 GEMM micro-benchmark (run on the GPU box from the repo root).
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/prof_p8
mkdir -p $OUT
export ONLY="${ONLY:-s3_pw}" CFGS="${CFGS:-60,64}" REPS=2 ROUNDS=2
CMD="python scripts/p8_bench.py"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --kernel-trace -d $OUT/pmc_a -o sp -- $CMD > $OUT/a.log 2> $OUT/a.err
rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA --kernel-trace -d $OUT/pmc_b -o sp -- $CMD > $OUT/b.log 2> $OUT/b.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/pmc_f -o sp -- $CMD > $OUT/f.log 2> $OUT/f.err
rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --kernel-trace -d $OUT/pmc_w -o sp -- $CMD > $OUT/w.log 2> $OUT/w.err
for d in pmc_a pmc_b pmc_f pmc_w; do
  db=$(ls $OUT/$d/*/*.db 2>/dev/null | head -1)
  [ -n "$db" ] && python scripts/rocpd_summary.py $db --pmc $db > $OUT/$d.txt 2>&1
done
tail -3 $OUT/a.err $OUT/b.err
cat $OUT/pmc_a.txt $OUT/pmc_b.txt $OUT/pmc_f.txt $OUT/pmc_w.txt | grep -v "^$" | grep -v "at::\|rocclr\|split_weights" | head -80
