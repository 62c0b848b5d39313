# usage (GPU box): bash tools/ab_run.sh <outdir> "<shape>;<shape>;..." <variant> ...
#   times the given tools/bench_one.py shapes with each build_ab/lib<variant>.so (PCC_BENCH_IMPL=0: AUTO dispatch)
OUT=gpurun_out/$1; shift
SHAPES=$1; shift
mkdir -p $OUT
for v in "$@"; do
  echo "== $v" >> $OUT/ab.log
  IFS=';' read -ra SH <<< "$SHAPES"
  for shape in "${SH[@]}"; do
    PCC_GEO_LIB=$PWD/build_ab/lib$v.so PCC_BENCH_IMPL=0 timeout 120 python tools/bench_one.py $shape 2>&1 | grep -v amdgpu.ids >> $OUT/ab.log
  done
done
cat $OUT/ab.log
