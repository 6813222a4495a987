# round-5 GPU call 4: 8-instruction GELU — full GPU suite, A/B against the round-4 form (second library)
mkdir -p gpurun_out/r05d
cd $GRAFT_REPO_ROOT
R4=$GRAFT_REPO_ROOT/wedetect_amd/libwedetect_hip_gelu_r4.so
Q="--steps 30 --warmup 8 --no-cpu-baseline --no-fp32-reference --no-host-fed --no-other-configs"
for i in 1 2; do
  WEDETECT_LIB=$R4 python bench.py $Q > gpurun_out/r05d/bench_r4gelu_$i.json 2> gpurun_out/r05d/bench_r4gelu_$i.err
  python bench.py $Q > gpurun_out/r05d/bench_new_$i.json 2> gpurun_out/r05d/bench_new_$i.err
done
export ROUNDS=4 REPS=6
{ echo "== round-4 GELU"; WEDETECT_LIB=$R4 ONLY=pw1 CFGS=64 python scripts/p8_bench.py; echo "== round-5 GELU"; ONLY=pw1 CFGS=64 python scripts/p8_bench.py; } > gpurun_out/r05d/p8_pw1.txt 2>&1
{ echo "== round-4 GELU"; WEDETECT_LIB=$R4 python scripts/mlp_fused_bench.py; echo "== round-5 GELU"; python scripts/mlp_fused_bench.py; } > gpurun_out/r05d/mlp_fused.txt 2>&1
( timeout 1200 python -m pytest tests -q -m gpu 2>&1 | tail -40 ) > gpurun_out/r05d/tests_full.log
for f in gpurun_out/r05d/*.json; do python - "$f" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    g=d['gemm_kernels']
    print(sys.argv[1].split('/')[-1], d['value'], d['unit'], d['ms_per_step'], d['roofline']['frac'], d['roofline'].get('avg_launch_us'), 'mlp128', g.get('fp16x3 fused block MLP 128x(128->512->128)/4w/dma',{}).get('avg_us'), 'mlp256', g.get('fp16x3 fused block MLP 128x(256->1024->256)/4w/frag',{}).get('avg_us'))
except Exception as e:
    print(sys.argv[1], 'FAILED', e)
PY
done
grep -v amdgpu.ids gpurun_out/r05d/p8_pw1.txt; grep -v amdgpu.ids gpurun_out/r05d/mlp_fused.txt | tail -20
tail -15 gpurun_out/r05d/tests_full.log
