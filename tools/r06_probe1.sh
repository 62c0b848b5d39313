# round 6 probe 1: fp16-split Winograd 16->16 kernel -- parity tests, then per-kernel time against the bf16 x 3 kernel on the same box
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r06p1; rm -rf $OUT; mkdir -p $OUT; export TMPDIR=/tmp
cd $R
timeout 900 python -m pytest tests/test_conv_gpu.py -x -q -m gpu -k "winograd or wino" > $OUT/pytest_wino.log 2>&1; echo "pytest rc $?" >> $OUT/pytest_wino.log
tail -15 $OUT/pytest_wino.log
cd /tmp
for v in f16s bf16; do
  if [ $v = bf16 ]; then export PCC_NO_F16S=1; else unset PCC_NO_F16S; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$v -o t -- timeout 300 env PCC_BENCH_IMPL=0 python $R/tools/bench_one.py 32 64 16 16 3 1 1 res > $OUT/bench_one_$v.log 2>&1
  echo "== $v"; tail -2 $OUT/bench_one_$v.log
  f=$(find $OUT/trace_$v -name "t_kernel_stats.csv" | head -1); head -6 $f | cut -c1-200
done
find $OUT -name "*agent_info.csv" -delete; find $OUT -name "*kernel_trace.csv" -size +3M -delete
