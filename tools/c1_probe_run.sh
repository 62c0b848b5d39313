# GPU box: timing probes of conv_cout1_mfma_kernel (-DPCC_C1_PROBE: 1 no P writes, 2 no gather reads, 4 no MFMA, 16 no plane loads), 16 -> 1 @64^3 x 32
for rep in 1 2; do for v in wbp0 c1p4 c1p1 c1p2 c1p16; do
  echo -n "$v: "; PCC_GEO_LIB=$PWD/build_ab/lib$v.so PCC_BENCH_IMPL=0 python tools/bench_one.py 32 64 16 1 3 1 1 2>&1 | grep impl | sed 's/.*tr1: //'
done; done
