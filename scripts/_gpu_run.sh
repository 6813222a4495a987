export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
timeout 300 python bench.py --no-cpu-baseline > gpurun_out/bench_fp16x3_d.json 2>/dev/null; cat gpurun_out/bench_fp16x3_d.json
