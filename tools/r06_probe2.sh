# round 6 probe 2: v_fma_mix split, fillers (default) vs block (mixk1), against bf16 x 3; per-kernel time from the kernel trace
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r06p2; rm -rf $OUT; mkdir -p $OUT; export TMPDIR=/tmp
cd $R
timeout 900 python -m pytest tests/test_conv_gpu.py -x -q -m gpu -k "winograd or wino" > $OUT/pytest_wino.log 2>&1; echo "pytest rc $?" >> $OUT/pytest_wino.log
tail -3 $OUT/pytest_wino.log
cd /tmp
for rep in 1 2; do
for v in f16s mixk1 bf16; do
  unset PCC_NO_F16S PCC_GEO_LIB
  [ $v = bf16 ] && export PCC_NO_F16S=1
  [ $v = mixk1 ] && export PCC_GEO_LIB=$R/build_ab/libmixk1.so
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_${v}_$rep -o t -- timeout 300 env PCC_BENCH_IMPL=0 python $R/tools/bench_one.py 32 64 16 16 3 1 1 res > $OUT/bench_one_${v}_$rep.log 2>&1
  f=$(find $OUT/trace_${v}_$rep -name "t_kernel_stats.csv" | head -1); echo "== $v $rep: $(grep conv16_wino $f | cut -d, -f2-8 | cut -c1-120)"
done; done
find $OUT -name "*agent_info.csv" -delete; find $OUT -name "*kernel_trace.csv" -delete
