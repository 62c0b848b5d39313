R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r06cfg2; rm -rf $OUT; mkdir -p $OUT; cd $R
timeout 900 python -m pytest tests/test_round6_gpu.py -x -q -m gpu 2>&1 | tail -15
python bench.py --workload configs2 --steps 10 --warmup 3 > $OUT/bench_configs2.log 2>&1; tail -1 $OUT/bench_configs2.log | cut -c1-1500
python tools/cli_wallclock.py > $OUT/cli_wallclock.md 2> $OUT/cli_wallclock.err; cat $OUT/cli_wallclock.md; tail -5 $OUT/cli_wallclock.err
python tools/cli_wallclock.py --fixed > $OUT/cli_wallclock_fixed.md 2>> $OUT/cli_wallclock.err; cat $OUT/cli_wallclock_fixed.md
