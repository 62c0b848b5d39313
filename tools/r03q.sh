for rep in 1 2; do
  python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('early', round(d['value']), round(d['ms_per_step'],3), round(d['config']['steady_state_ms_per_step'],3))"
  PCC_COPY_LATE=1 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('late ', round(d['value']), round(d['ms_per_step'],3), round(d['config']['steady_state_ms_per_step'],3))"
done
