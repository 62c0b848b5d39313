for rep in 1 2; do for v in wbp0 wbp1 wbp2 wbp4 wbp6; do
  echo -n "$v: "; PCC_GEO_LIB=$PWD/build_ab/lib$v.so PCC_BENCH_IMPL=0 python tools/bench_one.py 32 64 16 16 3 1 1 res 2>&1 | grep impl | sed 's/.*tr1: //'
done; done
for v in wbp0 wbp1 wbp2 wbp4 wbp6; do bash tools/pmc_shape.sh r05_wbp $v "32 64 16 16 3 1 1 res" PCC_GEO_LIB=$PWD/build_ab/lib$v.so 2>&1 | tail -1 | sed "s/^/$v /"; done
