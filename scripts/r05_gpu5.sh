# round-5 GPU call 5: (a) the fuzz tests after the reshape fix, (b) the N = 2 dry run of both bench modes on one GPU (gloo),
# (c) the two-steps-in-flight probe repeated
mkdir -p gpurun_out/r05e
cd $GRAFT_REPO_ROOT
( timeout 600 python -m pytest tests/test_gpu_fuzz.py tests/test_gpu_split.py -q -m gpu -k "random_shapes or persistent or p8_kernel" 2>&1 | tail -5 ) > gpurun_out/r05e/tests_fuzz.log
export WEDETECT_BENCH_SHARE_GPU=1 WEDETECT_BENCH_BACKEND=gloo
python bench.py --gpus 2 --batch 4 --steps 6 --warmup 2 > gpurun_out/r05e/bench_n2_dry.json 2> gpurun_out/r05e/bench_n2_dry.err
python bench.py --gpus 2 --mode retrieval --batch 4 --classes 20000 --steps 4 --warmup 1 > gpurun_out/r05e/bench_n2_retr_dry.json 2> gpurun_out/r05e/bench_n2_retr_dry.err
python bench.py --gpus 8 --batch 2 --steps 4 --warmup 1 > gpurun_out/r05e/bench_n8_dry.json 2> gpurun_out/r05e/bench_n8_dry.err
unset WEDETECT_BENCH_SHARE_GPU WEDETECT_BENCH_BACKEND
python scripts/two_stream_probe.py 30 > gpurun_out/r05e/two_stream.txt 2>&1
tail -3 gpurun_out/r05e/tests_fuzz.log
for f in bench_n2_dry bench_n2_retr_dry bench_n8_dry; do echo "== $f"; tail -c 1500 gpurun_out/r05e/$f.json; echo; tail -3 gpurun_out/r05e/$f.err; done
grep -v amdgpu gpurun_out/r05e/two_stream.txt
