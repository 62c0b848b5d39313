# GPU box: rocprofv3 kernel trace of bench.py (7 steps), conv kernels only, per (kernel, grid)
R=$(pwd); OUT=$R/gpurun_out/trace_bench; rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp; cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- timeout 300 python $R/bench.py --steps 6 --warmup 1 --no-cpu-baseline --no-secondary > $OUT/bench_profiled.log 2>&1
cd $R
python tools/summarize_trace.py $(find $OUT/trace -name t_kernel_trace.csv | head -1) 7 40 | grep -v "at::native\|rocclr" > $OUT/summary.md 2>&1
cat $OUT/summary.md
rm -rf $OUT/trace
