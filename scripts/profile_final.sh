#!/bin/bash
# rocprofv3 evidence for the current build (run on the GPU box from the repo root):
#   kernel trace + stats, one SQ counter pass, FETCH_SIZE and WRITE_SIZE passes (separate, as the
#   MI355X guide prescribes), all over the same `bench.py --steps 2 --warmup 1` command.
# usage: scripts/profile_final.sh [tag] [extra bench args]      -> gpurun_out/prof_<tag>/
export TMPDIR=/tmp
TAG=${1:-final}; shift
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p $OUT
# Round 6: per-kernel evidence (trace averages, SQ counters, per-launch HBM bytes) is taken on the SERIAL form of the step — one
# backbone chain, neck / head in line, the tile kernel forms of the pipelined backbone: what bench.py's serial leg measures (roofline.measured_in) — because in the default,
# pipelined form launches of several streams share the chip and a launch's duration / counters are not its own.  The default
# command gets a kernel trace of its own (summary_pipelined.txt): the whole picture, half-batch launches of the two image chains.
export WEDETECT_BB_CHAINS=1 WEDETECT_PIPE_NECK=0 WEDETECT_FORCE_TILE=1
CMD="python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-fp32-reference --no-host-fed --no-other-configs --no-calibrate $*"
timeout 900 rocprofv3 --kernel-trace --stats -d $OUT/trace -o p -- $CMD > $OUT/trace_bench.json 2> $OUT/trace.err
if [ -n "$TRACE_ONLY" ]; then     # kernel trace only (TRACE_ONLY=1): no counter passes
  python scripts/rocpd_summary.py $(ls $OUT/trace/*.db | head -1) > $OUT/summary.txt 2>&1
  rm -f $OUT/*/*.db; head -30 $OUT/summary.txt; cut -c1-300 $OUT/trace_bench.json; exit 0
fi
timeout 900 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE -d $OUT/pmc_sq -o p -- $CMD > /dev/null 2> $OUT/pmc_sq.err
timeout 900 rocprofv3 --pmc FETCH_SIZE -d $OUT/pmc_fetch -o p -- $CMD > /dev/null 2> $OUT/pmc_fetch.err
timeout 900 rocprofv3 --pmc WRITE_SIZE -d $OUT/pmc_write -o p -- $CMD > /dev/null 2> $OUT/pmc_write.err
T=$(ls $OUT/trace/*.db | head -1); S=$(ls $OUT/pmc_sq/*.db | head -1); F=$(ls $OUT/pmc_fetch/*.db | head -1); W=$(ls $OUT/pmc_write/*.db | head -1)
python scripts/rocpd_summary.py $T --pmc $S > $OUT/summary.txt 2>&1
python scripts/traffic_json.py $F $W > $OUT/traffic.json 2> $OUT/traffic.err
unset WEDETECT_BB_CHAINS WEDETECT_PIPE_NECK WEDETECT_FORCE_TILE
timeout 900 rocprofv3 --kernel-trace --stats -d $OUT/trace_pipe -o p -- $CMD > $OUT/trace_pipe_bench.json 2> $OUT/trace_pipe.err
python scripts/rocpd_summary.py $(ls $OUT/trace_pipe/*.db | head -1) > $OUT/summary_pipelined.txt 2>&1
rm -f $OUT/*/*.db            # keep the merge-back small: summaries only
head -40 $OUT/summary.txt; cat $OUT/trace_bench.json | cut -c1-400
