# PMC + trace of the fused-MLP microbench: bash scripts/mlp_pmc.sh  (MLP_PMC2=1: a second counter pass)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
export PYTHONPATH=.
OUT=gpurun_out/mlp_pmc; rm -rf $OUT; mkdir -p $OUT
CMD="python scripts/mlp_fused_bench.py 819200"
rocprofv3 --kernel-trace --stats -d $OUT/trace -o p -- $CMD > /dev/null 2> $OUT/trace.err
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE -d $OUT/pmc_sq -o p -- $CMD > /dev/null 2> $OUT/pmc_sq.err
T=$(ls $OUT/trace/*.db | head -1); S=$(ls $OUT/pmc_sq/*.db | head -1)
python scripts/rocpd_summary.py $T --pmc $S > $OUT/summary.txt 2>&1
if [ -n "$MLP_PMC2" ]; then
rocprofv3 --pmc SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA -d $OUT/pmc_sq2 -o p -- $CMD > /dev/null 2> $OUT/pmc_sq2.err
S2=$(ls $OUT/pmc_sq2/*.db | head -1)
python scripts/rocpd_summary.py $T --pmc $S2 > $OUT/summary2.txt 2>&1
fi
