#!/bin/bash
# PMC passes over the wide fused block-MLP micro-benchmark (run on the GPU box from the repo root): bash scripts/profile_mlpw.sh [c] [rows]
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/prof_mlpw
rm -rf $OUT; mkdir -p $OUT
CMD="python scripts/mlpw_bench.py ${1:-512} ${2:-51200}"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --kernel-trace -d $OUT/pmc_a -o sp -- $CMD > $OUT/a.log 2> $OUT/a.err
rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA --kernel-trace -d $OUT/pmc_b -o sp -- $CMD > $OUT/b.log 2> $OUT/b.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/pmc_f -o sp -- $CMD > $OUT/f.log 2> $OUT/f.err
rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --kernel-trace -d $OUT/pmc_w -o sp -- $CMD > $OUT/w.log 2> $OUT/w.err
for d in pmc_a pmc_b pmc_f pmc_w; do
  db=$(ls $OUT/$d/*.db $OUT/$d/*/*.db 2>/dev/null | head -1)
  [ -n "$db" ] && python scripts/rocpd_summary.py $db --pmc $db 2>&1 | grep -v "^$" | grep -v "at::\|rocclr\|split_weights\|layernorm\|Memcpy\|fill" > $OUT/$d.txt
done
cat $OUT/pmc_a.txt $OUT/pmc_b.txt $OUT/pmc_f.txt $OUT/pmc_w.txt
