# end-of-round evidence on the GPU box: WD_COMMIT=<commit of the build> [QUICK=1: profile + default bench only] bash scripts/final_evidence.sh
export WD_COMMIT=${WD_COMMIT:?set WD_COMMIT to the commit being measured (the box has no .git)}
mkdir -p gpurun_out/final
bash scripts/profile_final.sh r04 > gpurun_out/final/prof.log 2>&1
cp gpurun_out/prof_r04/traffic.json profiles/r04_traffic.json
python bench.py > gpurun_out/final/bench_latest.json 2> gpurun_out/final/bench_latest.err
if [ -z "$QUICK" ]; then
Q="--steps 30 --warmup 8 --no-cpu-baseline --no-fp32-reference --no-host-fed"
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
