# end-of-round evidence on the GPU box: WD_COMMIT=<commit of the build> [TAG=r05] [QUICK=1: profile + default bench only] bash scripts/final_evidence.sh
export WD_COMMIT=${WD_COMMIT:?set WD_COMMIT to the commit being measured (the box has no .git)}
TAG=${TAG:-r06}
export TMPDIR=/tmp
mkdir -p gpurun_out/final
bash scripts/profile_final.sh $TAG > gpurun_out/final/prof.log 2>&1
cp gpurun_out/prof_$TAG/traffic.json profiles/${TAG}_traffic.json
python bench.py > gpurun_out/final/bench_latest.json 2> gpurun_out/final/bench_latest.err
python bench.py --mode retrieval > gpurun_out/final/bench_retrieval_1m.json 2> gpurun_out/final/bench_retrieval_1m.err
# the retrieval kernel under the profiler: kernel trace + fabric traffic (separate PMC passes)
R="python bench.py --mode retrieval --steps 2 --warmup 1 --no-cpu-baseline"
OUT=$PWD/gpurun_out/prof_${TAG}_retr; mkdir -p $OUT
timeout 900 rocprofv3 --kernel-trace --stats -d $OUT/trace -o p -- $R > $OUT/trace_bench.json 2> $OUT/trace.err
timeout 900 rocprofv3 --pmc FETCH_SIZE -d $OUT/pmc_fetch -o p -- $R > /dev/null 2> $OUT/pmc_fetch.err
timeout 900 rocprofv3 --pmc WRITE_SIZE -d $OUT/pmc_write -o p -- $R > /dev/null 2> $OUT/pmc_write.err
python scripts/rocpd_summary.py $(ls $OUT/trace/*.db | head -1) > $OUT/summary.txt 2>&1
python scripts/traffic_json.py $(ls $OUT/pmc_fetch/*.db | head -1) $(ls $OUT/pmc_write/*.db | head -1) > $OUT/traffic.json 2> $OUT/traffic.err
rm -f $OUT/*/*.db
if [ -z "$QUICK" ]; then
Q="--steps 30 --warmup 8 --no-cpu-baseline --no-fp32-reference --no-host-fed --no-other-configs"
python bench.py $Q --batch 1 > gpurun_out/final/bench_base_b1.json 2>/dev/null
python bench.py $Q --batch 8 > gpurun_out/final/bench_base_b8.json 2>/dev/null
python bench.py $Q --batch 64 > gpurun_out/final/bench_base_b64.json 2>/dev/null
python bench.py $Q --arch tiny --batch 32 > gpurun_out/final/bench_tiny_b32_k80.json 2>/dev/null
python bench.py $Q --arch tiny --batch 1 > gpurun_out/final/bench_tiny_b1_k80.json 2>/dev/null
python bench.py $Q --arch large --batch 16 --classes 1203 > gpurun_out/final/bench_large_b16_k1203.json 2>/dev/null
python bench.py $Q --arch large --size 1280 --batch 4 --classes 1203 > gpurun_out/final/bench_large_1280_b4_k1203.json 2>/dev/null
python bench.py $Q --mode uni --classes 256 > gpurun_out/final/bench_uni_b32_k256.json 2>/dev/null
fi
for f in gpurun_out/final/bench_*.json; do python -c "
import json,sys
d=json.loads(open('$f').read().strip().splitlines()[-1]); print('$f'.split('/')[-1], d['value'], d['unit'], d['ms_per_step'], d['roofline']['frac'], d.get('sim_gemm',{}).get('frac'))"; done
