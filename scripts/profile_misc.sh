#!/bin/bash
# LDS bank-conflict counters of the depthwise 7x7 kernels and the similarity GEMM (run on the GPU box from the repo root)
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/prof_misc
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace -d $OUT/dw -o sp -- python scripts/dwconv_bench.py > $OUT/dw.log 2> $OUT/dw.err
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace -d $OUT/step -o sp -- python bench.py --steps 2 --warmup 1 --no-calibrate --no-cpu-baseline --no-fp32-reference --no-host-fed > $OUT/step.log 2> $OUT/step.err
for d in dw step; do
  db=$(ls $OUT/$d/*.db $OUT/$d/*/*.db 2>/dev/null | head -1)
  [ -n "$db" ] && python scripts/rocpd_summary.py $db --pmc $db 2>&1 | grep -v "^$" | grep -v "at::\|rocclr\|Memcpy\|fill" > $OUT/$d.txt
done
grep -i "dwconv\|kernel  " $OUT/dw.txt | cut -c1-260
grep -i "conv_gemm_kernel<1, 5\|dwconv\|layernorm\|kernel  " $OUT/step.txt | cut -c1-260
