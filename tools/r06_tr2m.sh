R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r06tr2m; rm -rf $OUT; mkdir -p $OUT; export TMPDIR=/tmp; cd $R
timeout 900 python -m pytest tests/test_conv_gpu.py tests/test_network_gpu.py -x -q -m gpu 2>&1 | tail -6
cd /tmp
for SHAPE in "32 32 32 16 3 2 1" "32 16 64 32 3 2 1"; do for rep in 1 2; do for v in f16s bf16; do
  unset PCC_NO_F16S; [ $v = bf16 ] && export PCC_NO_F16S=1
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/t_${v}_$rep -o t -- timeout 300 env PCC_BENCH_IMPL=0 python $R/tools/bench_one.py $SHAPE > $OUT/b_${v}_$rep.log 2>&1
  f=$(find $OUT/t_${v}_$rep -name "t_kernel_stats.csv" | head -1); echo "== $v $rep"; python - "$f" <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    if 'tr2' in r['Name'] or 'amax' in r['Name']: print(r['Name'][:60], r['Calls'], float(r['AverageNs'])/1e3, int(r['MinNs'])/1e3)
PY
done; done; done
unset PCC_NO_F16S; rm -rf $OUT/t_*
cd $R && python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-secondary --no-ab 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('bench', d['value'], d['ms_per_step'])"
