# round-5 GPU call 3: wide dwconv + LayerNorm kernel — identity tests, isolated timing, step A/B
mkdir -p gpurun_out/r05c
cd $GRAFT_REPO_ROOT
( timeout 600 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "dwconv7_ln" 2>&1 | tail -15 ) > gpurun_out/r05c/tests_dwln.log
python scripts/dwln_bench.py > gpurun_out/r05c/dwln_bench.txt 2>&1
Q="--steps 30 --warmup 8 --no-cpu-baseline --no-fp32-reference --no-host-fed --no-other-configs"
for i in 1 2; do
  python bench.py $Q > gpurun_out/r05c/bench_off_$i.json 2> gpurun_out/r05c/bench_off_$i.err
  WEDETECT_FUSE_DWLN_WIDE=256,512 python bench.py $Q > gpurun_out/r05c/bench_wide_$i.json 2> gpurun_out/r05c/bench_wide_$i.err
done
WEDETECT_FUSE_DWLN_WIDE=512 python bench.py $Q > gpurun_out/r05c/bench_wide512.json 2> gpurun_out/r05c/bench_wide512.err
tail -5 gpurun_out/r05c/tests_dwln.log; cat gpurun_out/r05c/dwln_bench.txt
for f in gpurun_out/r05c/*.json; do python - "$f" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1].split('/')[-1], d['value'], d['unit'], d['ms_per_step'], d['roofline']['frac'], d['roofline'].get('avg_launch_us'))
except Exception as e:
    print(sys.argv[1], 'FAILED', e)
PY
done
