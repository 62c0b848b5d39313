R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r06bench; mkdir -p $OUT; cd $R
python bench.py > $OUT/bench_default.log 2>&1; tail -1 $OUT/bench_default.log | cut -c1-300
