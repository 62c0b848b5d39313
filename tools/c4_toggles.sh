for i in 1 2 3 4 5 6; do python bench.py --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); s=d['secondary']; print(round(d['value']), round(d['ms_per_step'],3), d['config']['cpu_quota_throttled_periods_in_timed_region'], '|', round(s['value']), round(s['ms_per_step'],3), round(s['steady_ms_per_step'],3), s['cpu_quota_throttled_periods'])"; done
