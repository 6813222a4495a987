cd $GRAFT_REPO_ROOT
export TRACE_ONLY=1 WEDETECT_LN_FOLD=1
bash scripts/profile_final.sh fold > /dev/null 2>&1
head -24 gpurun_out/prof_fold/summary.txt | cut -c1-130
