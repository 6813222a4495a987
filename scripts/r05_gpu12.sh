# round-5 GPU call 12: the direct (no LDS transpose) hi/lo epilogue of the 256 x 256 kernel
mkdir -p gpurun_out/r05g
cd $GRAFT_REPO_ROOT
( WEDETECT_CSPLIT_DIRECT=1 timeout 600 python -m pytest tests/test_gpu_split.py -q -m gpu -k "p8_kernel or persistent or csplit or retrieval_on_the_256" 2>&1 | tail -4 ) > gpurun_out/r05g/tests_direct.log
export ROUNDS=4 REPS=6
{ echo "== through LDS (default)"; ONLY=pw1 CFGS=64 python scripts/p8_bench.py; echo "== direct (WEDETECT_CSPLIT_DIRECT=1)"; WEDETECT_CSPLIT_DIRECT=1 ONLY=pw1 CFGS=64 python scripts/p8_bench.py; } > gpurun_out/r05g/p8_pw1.txt 2>&1
Q="--steps 30 --warmup 8 --no-cpu-baseline --no-fp32-reference --no-host-fed --no-other-configs"
for i in 1 2; do
  python bench.py $Q > gpurun_out/r05g/bench_lds_$i.json 2> gpurun_out/r05g/bench_lds_$i.err
  WEDETECT_CSPLIT_DIRECT=1 python bench.py $Q > gpurun_out/r05g/bench_direct_$i.json 2> gpurun_out/r05g/bench_direct_$i.err
done
tail -3 gpurun_out/r05g/tests_direct.log; grep -v amdgpu gpurun_out/r05g/p8_pw1.txt
for f in gpurun_out/r05g/*.json; do python - "$f" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1].split('/')[-1], d['value'], d['unit'], d['ms_per_step'], d['roofline']['frac'], d['roofline'].get('avg_launch_us'))
except Exception as e:
    print(sys.argv[1], 'FAILED', e)
PY
done
