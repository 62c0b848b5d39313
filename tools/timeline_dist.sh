R=$(pwd); OUT=$R/gpurun_out/timeline_dist; rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp; cd /tmp
PCC_BENCH_FORCE_DIST=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29535 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -o t -- timeout 300 python $R/bench.py --steps 6 --warmup 1 --no-cpu-baseline --no-secondary > $OUT/bench_profiled.log 2>&1
cd $R
f=$(find $OUT/trace -name t_kernel_trace.csv | head -1)
python tools/timeline_gaps.py $f 4
python tools/timeline.py $f conv_cin1 4 75 | cut -c1-150
rm -rf $OUT/trace
