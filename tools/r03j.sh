R=$(pwd); OUT=$R/gpurun_out/r03j; rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp; cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- timeout 300 python $R/bench.py --steps 6 --warmup 1 --no-cpu-baseline --no-secondary > $OUT/bench_profiled.log 2>&1
cd $R
python tools/summarize_trace.py $(find $OUT/trace -name t_kernel_trace.csv | head -1) 7 30 > $OUT/summary.md 2>&1
cat $OUT/summary.md | head -45
find $OUT -name "*agent_info.csv" -delete; rm -f $(find $OUT/trace -name "t_kernel_trace.csv")
tools/pmc_shape.sh r03j tr2g_32_16 "32 32 32 16 3 2 1"
tools/pmc_shape.sh r03j tr2g_64_32 "32 16 64 32 3 2 1"
tools/pmc_shape.sh r03j cout1 "32 64 16 1 3 1 1"
