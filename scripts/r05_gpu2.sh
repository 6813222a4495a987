# round-5 GPU call 2: where does the pwconv1 epilogue's time go (arithmetic vs stores), and does taking the CUs out of phase help
mkdir -p gpurun_out/r05b
cd $GRAFT_REPO_ROOT
export ROUNDS=4 REPS=6
for s in 0 3 6 10; do
  echo "== WD_P8_STAGGER=$s (x ~4 us)"; WD_P8_STAGGER=$s ONLY=s3_pw CFGS=64 python scripts/p8_bench.py
done > gpurun_out/r05b/stagger.txt 2>&1
echo "== ablation build: 64 = full, 644 = no epilogue, 672 = arithmetic only (no stores), 704 = stores only (no arithmetic)" > gpurun_out/r05b/epi_abl.txt
WEDETECT_LIB=$GRAFT_REPO_ROOT/wedetect_amd/libwedetect_hip_abl.so ONLY=s3_pw1 CFGS=64,644,672,704 python scripts/p8_bench.py >> gpurun_out/r05b/epi_abl.txt 2>&1
Q="--steps 30 --warmup 8 --no-cpu-baseline --no-fp32-reference --no-host-fed --no-other-configs"
for i in 1 2; do
  python bench.py $Q > gpurun_out/r05b/bench_stag0_$i.json 2> gpurun_out/r05b/bench_stag0_$i.err
  WD_P8_STAGGER=6 python bench.py $Q > gpurun_out/r05b/bench_stag6_$i.json 2> gpurun_out/r05b/bench_stag6_$i.err
done
cat gpurun_out/r05b/stagger.txt gpurun_out/r05b/epi_abl.txt
for f in gpurun_out/r05b/*.json; do python - "$f" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1].split('/')[-1], d['value'], d['unit'], d['ms_per_step'], d['roofline']['frac'], d['roofline'].get('avg_launch_us'))
except Exception as e:
    print(sys.argv[1], 'FAILED', e)
PY
done
