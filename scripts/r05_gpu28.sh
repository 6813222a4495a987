cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/final
Q="--steps 50 --warmup 10 --no-cpu-baseline --no-fp32-reference --no-host-fed --no-other-configs"
for i in 1 2; do
for f in 1024 512 4096; do
WD_P8_PERSIST_MINK=$f python bench.py $Q 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('persist_mink $f', d['value'], d['ms_per_step'], d['roofline']['avg_launch_us'])"
done; done | tee gpurun_out/final/ab_persist_mink.txt
