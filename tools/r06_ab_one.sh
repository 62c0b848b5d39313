# same-box A/B of one layer shape between library builds, per-kernel time from the kernel trace:
#   tools/r06_ab_one.sh <tag> "<bench_one args>" <kernel substring> lib1 lib2 ...   (lib = name in build_ab/ or "tree")
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=$1; ARGS=$2; KSUB=$3; shift 3
OUT=$R/gpurun_out/$TAG; rm -rf $OUT; mkdir -p $OUT; export TMPDIR=/tmp
cd /tmp
for rep in 1 2 3; do
for v in "$@"; do
  unset PCC_GEO_LIB
  [ $v != tree ] && export PCC_GEO_LIB=$R/build_ab/lib$v.so
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/t_${v}_$rep -o t -- timeout 300 env PCC_BENCH_IMPL=0 python $R/tools/bench_one.py $ARGS > $OUT/b_${v}_$rep.log 2>&1
  f=$(find $OUT/t_${v}_$rep -name "t_kernel_stats.csv" | head -1)
  echo "== $v $rep: $(grep "$KSUB" $f | head -1 | awk -F'",' '{print $2}' | cut -d, -f1-7)"
done; done
rm -rf $OUT/t_*
