# GPU box: A/B of two source trees on the same box: build_ab/old (a `git archive` of an earlier commit, built) vs the working tree.
# Alternates runs to cancel drift; prints blocks/s, ms/step, steady ms/step per run.
R=$(pwd)
run() { (cd $1 && python bench.py --steps ${STEPS:-40} --warmup 5 --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$2', round(d['value']), round(d['ms_per_step'],3), round(d['config'].get('steady_state_ms_per_step',0),3), round(d['roofline']['avg_launch_ms'],4))"); }
for i in 1 2 3; do run $R/build_ab/old old; run $R new; done
