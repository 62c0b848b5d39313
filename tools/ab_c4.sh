# GPU box: alternating A/B of the configs[4] bench (fp16 mode): current dispatch vs PCC_NO_TR2M=1 (tiled stride-2 transposed kernels); "$@" = extra bench flags
for i in 1 2 3; do for v in "" 1; do
  if [ -n "$v" ]; then export PCC_NO_TR2M=1; else unset PCC_NO_TR2M; fi
  python bench.py --workload configs4 --steps 80 --warmup 10 "$@" 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); c=d['config']; print('NO_TR2M=$v', round(d['value']), '128^3 blocks/s', round(d['ms_per_step'],3), 'ms/step; steady', round(c['steady_state_ms_per_step'],3), 'cores busy', c['host_cores_busy_per_rank'], 'throttled', c['cpu_quota_throttled_periods_in_timed_region'], 'coder threads', c['coder_threads_per_rank'])"
done; done
