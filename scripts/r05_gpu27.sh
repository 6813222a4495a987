cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
Q="--steps 3 --warmup 1 --no-cpu-baseline --no-fp32-reference --no-host-fed --no-other-configs"
for cfg in "large:--arch large --batch 16 --classes 1203" "uni:--mode uni --classes 256"; do
  name=${cfg%%:*}; args=${cfg#*:}
  OUT=$PWD/gpurun_out/prof_r05_$name; mkdir -p $OUT
  timeout 400 rocprofv3 --kernel-trace --stats -d $OUT/trace -o p -- python bench.py $Q $args > $OUT/trace_bench.json 2> $OUT/trace.err
  python scripts/rocpd_summary.py $(ls $OUT/trace/*.db | head -1) > $OUT/summary.txt 2>&1
  rm -f $OUT/*/*.db
  head -12 $OUT/summary.txt | cut -c1-120
done
