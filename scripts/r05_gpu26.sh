cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/final
( timeout 900 python -m pytest tests/test_gpu_network.py tests/test_gpu_configs.py -q -x 2>&1 | tail -8 ) > gpurun_out/final/gputest_net.log
tail -4 gpurun_out/final/gputest_net.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
