cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/variants
rm -f gpurun_out/parity_r05.jsonl
WEDETECT_LN_FOLD=1 timeout 1200 python -m pytest tests/test_gpu_network.py tests/test_gpu_configs.py tests/test_gpu_detector.py tests/test_gpu_precision.py tests/test_gpu_entry.py -q -m gpu 2>&1 | tail -12 > gpurun_out/variants/tests_lnfold.log
cp gpurun_out/parity_r05.jsonl gpurun_out/variants/parity_lnfold.jsonl
cat gpurun_out/variants/tests_lnfold.log | cut -c1-250
