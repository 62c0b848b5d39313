R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r04a; rm -rf $OUT; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- timeout 180 python $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-secondary > $OUT/bench_profiled.log 2>&1
find $OUT -name "*agent_info.csv" -delete
python $R/tools/summarize_trace.py $(find $OUT/trace -name "t_kernel_trace.csv") 6 34
