# fp16 mode (configs[4]): the fp16 tests, then the mode's bench line three times; PCC_NO_SPLIT_DIRECT=1 = the routing before round 6 (A/B)
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r06fp16; rm -rf $OUT; mkdir -p $OUT; cd $R
timeout 1200 python -m pytest tests/test_conv_gpu.py tests/test_network_gpu.py tests/test_codec_gpu.py -x -q -m gpu -k "f16 or fp16 or config4 or 128" 2>&1 | tail -4
for rep in 1 2 3; do for v in new old; do
  unset PCC_NO_SPLIT_DIRECT; [ $v = old ] && export PCC_NO_SPLIT_DIRECT=1
  python bench.py --workload configs4 --steps 40 --warmup 5 2>/dev/null | tail -1 > $OUT/b_${v}_$rep.json
  python -c "import json; d=json.load(open('$OUT/b_${v}_$rep.json')); print('$v $rep', round(d['value'],1), round(d['ms_per_step'],3), round(d.get('steady_ms_per_step',0),3))"
done; done
