R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r06fp16t; rm -rf $OUT; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/t -o t -- python $R/bench.py --workload configs4 --steps 30 --warmup 5 > $OUT/bench.log 2>&1
tail -1 $OUT/bench.log | cut -c1-300
ls -la $(find $OUT/t -name "*kernel_trace.csv")
f=$(ls -S $(find $OUT/t -name "*kernel_trace.csv") | head -1)
python $R/tools/trace_busy.py $f 0.3
head -1 $f
cp $f $OUT/kernel_trace.csv; rm -rf $OUT/t
