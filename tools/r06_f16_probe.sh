# conv_f16_kernel<16> @128^3 x 8 (fp16 mode's dominant kernel) alone, library builds side by side: kernel time from the trace
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r06f16p; rm -rf $OUT; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
for rep in 1 2; do for v in "$@"; do for args in "16 8 128 res" "16 8 128"; do
  unset PCC_GEO_LIB; [ $v != tree ] && export PCC_GEO_LIB=$R/build_ab/lib$v.so
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/t -o t -- timeout 300 python $R/tools/bench_f16.py $args > $OUT/b.log 2>&1
  f=$(find $OUT/t -name "t_kernel_stats.csv" | head -1)
  echo "== $v [$args] $rep: $(grep "conv_f16" $f | head -1 | awk -F'",' '{print $2}' | cut -d, -f1-6)"
  rm -rf $OUT/t
done; done; done
