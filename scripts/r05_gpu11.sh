cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/final gpurun_out/variants
( WEDETECT_FUSE_DWLN_WIDE=256,384,512 timeout 600 python -m pytest tests/test_gpu_network.py tests/test_gpu_split.py -q -m gpu -k "hipgraph or retrieval_on_the_256" 2>&1 | tail -3 ) > gpurun_out/variants/tests_fix.log
export WD_COMMIT=2e8ffe7 TAG=r05 QUICK=1
bash scripts/final_evidence.sh > gpurun_out/final/evidence_quick.log 2>&1
cat gpurun_out/variants/tests_fix.log; tail -4 gpurun_out/final/evidence_quick.log
