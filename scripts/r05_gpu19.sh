cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/final
( timeout 900 python -m pytest tests/test_gpu_entry.py -q -x 2>&1 | tail -25 ) > gpurun_out/final/gputest_entry.log
tail -5 gpurun_out/final/gputest_entry.log
timeout 600 python bench.py > gpurun_out/final/bench_latest_box_c.json 2> gpurun_out/final/bench_latest_box_c.err
python -c "
import json
d=json.loads(open('gpurun_out/final/bench_latest_box_c.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['avg_launch_us'], d['roofline']['effective_mhz'])"
# SQ counters of the retrieval launch (own pass, no trace domains beside --kernel-trace)
R="python bench.py --mode retrieval --steps 2 --warmup 1 --no-cpu-baseline"
OUT=$PWD/gpurun_out/prof_r05_retr_sq; mkdir -p $OUT
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --kernel-trace -d $OUT/pmc_sq -o p -- $R > /dev/null 2> $OUT/pmc_sq.err
S=$(ls $OUT/pmc_sq/*.db | head -1)
python scripts/rocpd_summary.py $S --pmc $S > $OUT/summary.txt 2>&1
rm -f $OUT/*/*.db
head -30 $OUT/summary.txt | cut -c1-300
