cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/final
( timeout 1500 python -m pytest tests -q -m gpu 2>&1 | tail -25 ) > gpurun_out/final/gputest_full.log
tail -3 gpurun_out/final/gputest_full.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/final/smoke.log 2>&1; tail -1 gpurun_out/final/smoke.log
timeout 900 python bench.py > gpurun_out/final/bench_latest_final_commit.json 2> gpurun_out/final/bench_latest_final_commit.err
wc -l gpurun_out/final/bench_latest_final_commit.json
python -c "
import json
d=json.loads(open('gpurun_out/final/bench_latest_final_commit.json').read().strip().splitlines()[-1]); r=d['roofline']; print(d['value'], d['ms_per_step'], r['frac'], r['avg_launch_us'], r['effective_mhz'], r['traffic_provenance']['taken_on_these_sources'], [ (k[:12], v['value']) for k,v in d['other_configs'].items()])"
