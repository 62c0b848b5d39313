# GPU box: bench.py twice per setting, alternating: tools/ab_env.sh "VAR=val ..." ["VAR2=val ..."]  (first setting: defaults)
run() {
  tag=$1; shift
  env "$@" python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$tag', round(d['value']), round(d['ms_per_step'],3), round(d['config']['steady_state_ms_per_step'],3))"
}
for rep in 1 2; do
  run default X=1
  for s in "$@"; do run "$s" $s; done
done
