cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/final
# the launcher path after the stdout change: two gloo ranks time-slicing the one GPU; stdout must hold exactly one line
WEDETECT_BENCH_SHARE_GPU=1 WEDETECT_BENCH_BACKEND=gloo timeout 600 python bench.py --gpus 2 --steps 5 --warmup 2 > gpurun_out/final/bench_n2_dry.json 2> gpurun_out/final/bench_n2_dry.err
echo rc=$?; wc -l gpurun_out/final/bench_n2_dry.json
python -c "
import json
d=json.loads(open('gpurun_out/final/bench_n2_dry.json').read().strip().splitlines()[-1]); print(d['value'], d['n_gpus'], d['ms_per_step'], d['ranks_seen'], d['collective_backend'], d['per_rank'])"
WEDETECT_BENCH_SHARE_GPU=1 WEDETECT_BENCH_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 5 --warmup 2 --mode retrieval --classes 100000 > gpurun_out/final/bench_n2_dry_retr.json 2> gpurun_out/final/bench_n2_dry_retr.err
echo rc=$?; wc -l gpurun_out/final/bench_n2_dry_retr.json; cut -c1-600 gpurun_out/final/bench_n2_dry_retr.json
