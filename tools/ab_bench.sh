# GPU box: same-box A/B of bench.py -- the current kernels against the round-2 paths (PCC_NO_TR2M=1 PCC_WINO_PER_GROUP=1), two repetitions
mkdir -p gpurun_out/ab_bench
for rep in 1 2; do
  python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('new ', round(d['value']), round(d['ms_per_step'],3), d['config']['steady_state_ms_per_step'], round(d['roofline']['avg_launch_ms'],4))"
  PCC_NO_TR2M=1 PCC_WINO_PER_GROUP=1 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('old ', round(d['value']), round(d['ms_per_step'],3), d['config']['steady_state_ms_per_step'], round(d['roofline']['avg_launch_ms'],4))"
done
