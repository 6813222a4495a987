mkdir -p gpurun_out/r05f
cd $GRAFT_REPO_ROOT
python scripts/latency_small_batch.py > gpurun_out/r05f/small_batch.txt 2>&1
grep -v amdgpu gpurun_out/r05f/small_batch.txt
