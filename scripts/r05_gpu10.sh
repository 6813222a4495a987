# round-5 last GPU call: the network / configuration tests under the switches that select OTHER kernels or schedules, then the
# evidence of the final commit (profile + default bench + retrieval bench)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/variants
T="tests/test_gpu_network.py tests/test_gpu_configs.py tests/test_gpu_detector.py tests/test_gpu_evaluate.py"
WEDETECT_DAG=0 timeout 900 python -m pytest $T -q -m gpu 2>&1 | tail -3 > gpurun_out/variants/tests_dag0.log; cp gpurun_out/parity_r05.jsonl gpurun_out/variants/parity_dag0.jsonl; rm -f gpurun_out/parity_r05.jsonl
WEDETECT_FUSE_DWLN_WIDE=256,384,512 WEDETECT_RETR_P8=0 WEDETECT_RETRIEVAL_PRECISION=fp32 timeout 900 python -m pytest $T -q -m gpu 2>&1 | tail -3 > gpurun_out/variants/tests_dwlnwide_retr_old.log; cp gpurun_out/parity_r05.jsonl gpurun_out/variants/parity_dwlnwide.jsonl; rm -f gpurun_out/parity_r05.jsonl
export WD_COMMIT=2320c99 TAG=r05 QUICK=1
bash scripts/final_evidence.sh > gpurun_out/final/evidence_quick.log 2>&1
cat gpurun_out/variants/tests_dag0.log gpurun_out/variants/tests_dwlnwide_retr_old.log; tail -4 gpurun_out/final/evidence_quick.log
