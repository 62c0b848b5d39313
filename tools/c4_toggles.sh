run() { echo "== $1 $2"; env $1 python bench.py $2 --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['ms_per_step'],3), d['config'].get('steady_state_ms_per_step'))"; }
run A=1 "--workload configs4 --steps 20 --warmup 3"
run A=1 "--workload configs4 --steps 20 --warmup 3"
run A=1 ""
run A=1 ""
