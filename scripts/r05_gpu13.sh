cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/final gpurun_out/variants
( timeout 600 python -m pytest tests/test_gpu_split.py tests/test_gpu_network.py -q -m gpu -k "retrieval_on_the_256 or hipgraph or bank_scorer" 2>&1 | tail -3 ) > gpurun_out/variants/tests_fix2.log
export WD_COMMIT=af23c9f TAG=r05 QUICK=1
bash scripts/final_evidence.sh > gpurun_out/final/evidence_quick.log 2>&1
cat gpurun_out/variants/tests_fix2.log; tail -4 gpurun_out/final/evidence_quick.log
