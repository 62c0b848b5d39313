# same-box kernel time of the 16 -> 16 @64^3 + residual launch for several library builds (build_ab/lib<name>.so or `tree`): the probe table of tools/ubench/conv_dir_f16s.hip.inc
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r06dirp; rm -rf $OUT; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
for rep in 1 2; do for v in "$@"; do
  unset PCC_GEO_LIB; [ $v != tree ] && export PCC_GEO_LIB=$R/build_ab/lib$v.so
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/t -o t -- timeout 300 env PCC_BENCH_IMPL=0 python $R/tools/bench_one.py 32 64 16 16 3 1 1 res > $OUT/b.log 2>&1
  f=$(find $OUT/t -name "t_kernel_stats.csv" | head -1)
  echo "== $v $rep: $(grep "conv16" $f | head -1 | awk -F'",' '{print $2}' | cut -d, -f1-6)"
  rm -rf $OUT/t
done; done
