export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_gpu_split.py -q -x 2>&1 | tail -4
timeout 300 python scripts/presplit_bench.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/presplit_ab.log
