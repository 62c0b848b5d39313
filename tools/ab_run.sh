# usage (GPU box): bash tools/ab_run.sh <outdir> <variant> ...   -- times the three Winograd layer shapes with each build_ab/lib<variant>.so
OUT=gpurun_out/$1; shift
mkdir -p $OUT
for v in "$@"; do
  echo "== $v" >> $OUT/ab.log
  for shape in "32 64 16 16 3 1 1 res" "32 64 16 16 3 1 1" "32 32 32 32 3 1 1 res" "32 16 64 64 3 1 1 res"; do
    PCC_GEO_LIB=$PWD/build_ab/lib$v.so PCC_BENCH_IMPL=0 python tools/bench_one.py $shape 2>&1 | grep -v amdgpu.ids >> $OUT/ab.log
  done
done
cat $OUT/ab.log
