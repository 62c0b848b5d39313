# the driver's own command, repeated on one fresh box, with the arrival spacing of the chunks (PCC_BENCH_STAMPS) -- how often does a 20-step run stall?
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out/r06drv
for i in ${RUNS:-1 2 3 4 5}; do
  PCC_BENCH_STAMPS=1 python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r06drv/run$i.log 2> gpurun_out/r06drv/run$i.err
  python3 - gpurun_out/r06drv/run$i.log gpurun_out/r06drv/run$i.err <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
st=[l for l in open(sys.argv[2]) if l.startswith('stamps')]
print(round(d['value']), round(d['ms_per_step'],3), 'steady', round(d['config']['steady_state_ms_per_step'],3), 'sec', round(d['secondary']['value']), '| stamps', st[0].strip()[:200] if st else '')
PY
done
