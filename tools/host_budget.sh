#!/bin/bash
# GPU box: blocks/s of the 1-GPU bench against the host cores one rank may use (taskset), and 4 ranks of 4 cores sharing the GPU --
# the host budget an 8-rank node has to provide (table in DESIGN_HISTORY.md section 6).   tools/host_budget.sh [steps]
STEPS=${1:-100}
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
one() { python - "$1" <<'PY'
import json, sys
l = [x for x in open(sys.argv[1]).read().splitlines() if x.startswith('{')]
d = json.loads(l[-1]); c = d['config']
print(f"{d['value']:8.0f} blocks/s  {d['ms_per_step']:.3f} ms/step  steady {c['steady_state_ms_per_step']:.3f}  host busy {c['host_cores_busy_per_rank']:.2f} of {c['host_cores_per_rank']} cores  coder threads {c['coder_threads_per_rank']}  host_bound {c['host_bound']}  throttled periods {c['cpu_quota_throttled_periods_in_timed_region']}")
PY
}
for n in 16 8 6 4 3 2 1; do
  taskset -c 0-$((n - 1)) python bench.py --steps $STEPS --no-cpu-baseline --no-secondary > /tmp/hb_$n.log 2>&1
  printf "cores/rank %2d: " $n; one /tmp/hb_$n.log
done
echo "4 processes x 4 cores sharing the one GPU (aggregate = the sum):"
for r in 0 1 2 3; do
  taskset -c $((4 * r))-$((4 * r + 3)) python bench.py --steps $STEPS --no-cpu-baseline --no-secondary > /tmp/hb_s$r.log 2>&1 &
done
wait
for r in 0 1 2 3; do printf "  proc %d: " $r; one /tmp/hb_s$r.log; done
echo "8 processes x 2 cores sharing the one GPU:"
for r in 0 1 2 3 4 5 6 7; do
  taskset -c $((2 * r))-$((2 * r + 1)) python bench.py --steps $((STEPS / 2)) --no-cpu-baseline --no-secondary > /tmp/hb_e$r.log 2>&1 &
done
wait
for r in 0 1 2 3 4 5 6 7; do printf "  proc %d: " $r; one /tmp/hb_e$r.log; done
