# GPU box: the 1-GPU bench with and without an initialised RCCL process group (world 1)
one() { python bench.py --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', round(d['value']), round(d['ms_per_step'],3), d['roofline']['avg_launch_ms'], d['config']['host_cores_busy_per_rank'])"; }
for i in 1 2; do
one plain
PCC_BENCH_FORCE_DIST=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=2953$i RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 one dist
done
