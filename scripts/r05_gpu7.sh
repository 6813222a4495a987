cd $GRAFT_REPO_ROOT
export WD_COMMIT=2fa13c2 TAG=r05
mkdir -p gpurun_out/final
( timeout 1500 python -m pytest tests -q -m gpu 2>&1 | tail -25 ) > gpurun_out/final/gputest_full.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/final/smoke.log 2>&1
bash scripts/final_evidence.sh > gpurun_out/final/evidence.log 2>&1
tail -4 gpurun_out/final/gputest_full.log; cat gpurun_out/final/smoke.log | tail -2; tail -16 gpurun_out/final/evidence.log
