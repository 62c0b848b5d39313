run() { PCC_GEO_LIB=$PWD/build_ab/lib$1.so python bench.py --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', round(d['value']), round(d['ms_per_step'],4))"; }
for i in 1 2 3; do run wold; run wnew; done
