R=$(pwd); OUT=$R/gpurun_out/timeline; rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp; cd /tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -o t -- timeout 300 python $R/bench.py --steps 6 --warmup 1 --no-cpu-baseline --no-secondary > $OUT/bench_profiled.log 2>&1
cd $R
f=$(find $OUT/trace -name t_kernel_trace.csv | head -1)
python tools/timeline.py $f conv_cin1 3 140 > $OUT/timeline.txt
m=$(find $OUT/trace -name t_memory_copy_trace.csv | head -1); [ -n "$m" ] && cp $m $OUT/memcopy.csv
cp $f $OUT/kernel_trace.csv
rm -rf $OUT/trace
cat $OUT/timeline.txt
