# round 6 probe 3: full GPU suite with the fp16-split 16-channel layers + amax side channel, then bench line and a kernel trace of it
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r06p6; rm -rf $OUT; mkdir -p $OUT; export TMPDIR=/tmp
cd $R
timeout 1500 python -m pytest tests -x -q -m gpu > $OUT/pytest_gpu.log 2>&1; echo "pytest rc $?" >> $OUT/pytest_gpu.log
tail -8 $OUT/pytest_gpu.log
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --no-ab > $OUT/bench.log 2>&1; tail -1 $OUT/bench.log | cut -c1-400
PCC_NO_F16S=1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --no-ab > $OUT/bench_nof16s.log 2>&1; tail -1 $OUT/bench_nof16s.log | cut -c1-200
cd /tmp
PCC_BENCH_NO_PRIME=1 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- timeout 180 python $R/bench.py --steps 10 --warmup 5 --no-cpu-baseline --no-secondary --no-ab > $OUT/bench_profiled.log 2>&1
f=$(find $OUT/trace -name "t_kernel_stats.csv" | head -1); head -25 $f | cut -d, -f1-4 | cut -c1-150
find $OUT -name "*agent_info.csv" -delete
