# round-5 GPU call 1: new parity tests + A/B of the DAG schedule, retrieval kernels, two-stream probe
mkdir -p gpurun_out/r05a
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_split.py tests/test_gpu_network.py tests/test_gpu_entry.py tests/test_gpu_evaluate.py tests/test_gpu_precision.py -q -m gpu -k "heavy_tailed or retrieval or split_weights or bank_scorer or dag or tiny_config0 or reference_goldens or config0 or device_retrieval or pipelined" 2>&1 | tail -25 ) > gpurun_out/r05a/tests_new.log
Q="--steps 30 --warmup 8 --no-cpu-baseline --no-fp32-reference --no-host-fed --no-other-configs"
for i in 1 2; do
  WEDETECT_DAG=0 python bench.py $Q > gpurun_out/r05a/bench_dag0_$i.json 2> gpurun_out/r05a/bench_dag0_$i.err
  python bench.py $Q > gpurun_out/r05a/bench_dag1_$i.json 2> gpurun_out/r05a/bench_dag1_$i.err
done
WEDETECT_RETR_P8=0 python bench.py --mode retrieval --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r05a/retr_old.json 2> gpurun_out/r05a/retr_old.err
python bench.py --mode retrieval --steps 10 --warmup 3 > gpurun_out/r05a/retr_p8.json 2> gpurun_out/r05a/retr_p8.err
python scripts/two_stream_probe.py 20 > gpurun_out/r05a/two_stream.txt 2>&1
python bench.py --steps 20 --warmup 5 > gpurun_out/r05a/bench_full.json 2> gpurun_out/r05a/bench_full.err
for f in gpurun_out/r05a/*.json; do python - "$f" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1].split('/')[-1], d['value'], d['unit'], d['ms_per_step'], d['roofline']['frac'], d['roofline'].get('avg_launch_us'))
except Exception as e:
    print(sys.argv[1], 'FAILED', e)
PY
done
cat gpurun_out/r05a/tests_new.log | tail -8
cat gpurun_out/r05a/two_stream.txt
