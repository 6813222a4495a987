mkdir -p gpurun_out/r05h
cd $GRAFT_REPO_ROOT
( timeout 600 python -m pytest tests/test_gpu_split.py tests/test_gpu_network.py -q -m gpu -k "layernorm_fold" 2>&1 | tail -30 ) > gpurun_out/r05h/tests_fold.log
Q="--steps 30 --warmup 8 --no-cpu-baseline --no-fp32-reference --no-host-fed --no-other-configs"
for i in 1 2; do
  python bench.py $Q > gpurun_out/r05h/bench_off_$i.json 2> gpurun_out/r05h/bench_off_$i.err
  WEDETECT_LN_FOLD=1 python bench.py $Q > gpurun_out/r05h/bench_fold_$i.json 2> gpurun_out/r05h/bench_fold_$i.err
done
tail -30 gpurun_out/r05h/tests_fold.log | cut -c1-220
for f in gpurun_out/r05h/*.json; do python - "$f" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1].split('/')[-1], d['value'], d['unit'], d['ms_per_step'], d['roofline']['frac'], d['roofline'].get('avg_launch_us'), d['config']['fp16x3_range_guard_tripped'])
except Exception as e:
    print(sys.argv[1], 'FAILED', e); print(open(sys.argv[1].replace('.json','.err')).read()[-800:])
PY
done
