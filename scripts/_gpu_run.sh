export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 300 python bench.py > gpurun_out/bench_final.json 2>/dev/null; cut -c1-300 gpurun_out/bench_final.json; echo
bash scripts/profile_final.sh fp16x3 > gpurun_out/prof_fp16x3.log 2>&1; tail -42 gpurun_out/prof_fp16x3.log | head -30
python -c "
import __graft_entry__ as g
g.smoke()
" 2>&1 | tail -3
