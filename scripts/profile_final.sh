#!/bin/bash
# rocprofv3 evidence for the final round-1 build (run on the GPU box from the repo root).
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/prof_final
mkdir -p $OUT
CMD="python bench.py --steps 2 --warmup 1 --no-cpu-baseline"
rocprofv3 --kernel-trace --stats -d $OUT/trace -o r01f -- $CMD > $OUT/trace_bench.json 2> $OUT/trace.err
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE -d $OUT/pmc_sq -o r01f -- $CMD > /dev/null 2> $OUT/pmc_sq.err
rocprofv3 --pmc FETCH_SIZE -d $OUT/pmc_fetch -o r01f -- $CMD > /dev/null 2> $OUT/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE -d $OUT/pmc_write -o r01f -- $CMD > /dev/null 2> $OUT/pmc_write.err
ls $OUT/*/
