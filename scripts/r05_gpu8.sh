cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/dbg
timeout 1500 python -X faulthandler -m pytest tests -v -m gpu -p no:cacheprovider > gpurun_out/dbg/full_v.log 2>&1
rc=$?
echo "rc=$rc"
grep -n "Memory access fault\|core dumped\|Fatal Python\|Aborted\|HSA_STATUS" gpurun_out/dbg/full_v.log | head
grep -n "PASSED\|FAILED\|ERROR" gpurun_out/dbg/full_v.log | tail -3
tail -5 gpurun_out/dbg/full_v.log | cut -c1-300
if [ $rc -ne 0 ]; then
  WEDETECT_DAG=0 timeout 1500 python -X faulthandler -m pytest tests -q -m gpu -p no:cacheprovider > gpurun_out/dbg/full_dag0.log 2>&1
  echo "dag0 rc=$?"; tail -3 gpurun_out/dbg/full_dag0.log | cut -c1-300
fi
