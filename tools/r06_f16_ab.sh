# conv_f16_kernel<16> alone (tools/bench_f16.py) and the fp16-mode line, residual look-ahead 6 (tree) against 3 (build_ab/libresd3.so)
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r06f16ab; rm -rf $OUT; mkdir -p $OUT; export TMPDIR=/tmp; cd $R
timeout 1200 python -m pytest tests/test_conv_gpu.py tests/test_network_gpu.py tests/test_codec_gpu.py -x -q -m gpu -k "f16 or fp16 or config4 or 128" 2>&1 | tail -3
for rep in 1 2 3; do for v in tree resd3; do
  unset PCC_GEO_LIB; [ $v != tree ] && export PCC_GEO_LIB=$R/build_ab/lib$v.so
  echo "$v $rep: $(python tools/bench_f16.py 16 8 128 res 2>/dev/null | tail -1 | cut -c1-200)"
  echo "$v $rep nores: $(python tools/bench_f16.py 16 8 128 2>/dev/null | tail -1 | cut -c1-200)"
  python bench.py --workload configs4 --steps 40 --warmup 5 2>/dev/null | tail -1 > $OUT/b_${v}_$rep.json
  python -c "import json; d=json.load(open('$OUT/b_${v}_$rep.json')); print('$v $rep configs4', round(d['value'],1), round(d['ms_per_step'],3))"
done; done
