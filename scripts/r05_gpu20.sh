cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/final
( timeout 900 python -m pytest tests/test_gpu_entry.py -q 2>&1 | tail -25 ) > gpurun_out/final/gputest_entry.log
tail -8 gpurun_out/final/gputest_entry.log
timeout 600 python bench.py --mode retrieval > gpurun_out/final/bench_retrieval_1m_box_c.json 2> gpurun_out/final/bench_retrieval_1m_box_c.err
wc -l gpurun_out/final/bench_retrieval_1m_box_c.json
python -c "
import json
d=json.loads(open('gpurun_out/final/bench_retrieval_1m_box_c.json').read().strip().splitlines()[-1]); r=d['roofline']; print(d['value'], d['ms_per_step'], r['frac'], r.get('effective_mhz'), r.get('frac_at_effective_clock'), r.get('effective_mhz_min_max'))"
