mkdir -p gpurun_out/r03b
python -m pytest tests/test_conv_gpu.py -m gpu -x -q -k "winograd" > gpurun_out/r03b/tests.log 2>&1; tail -5 gpurun_out/r03b/tests.log
python tools/stress_wino.py 7 80 > gpurun_out/r03b/stress.log 2>&1; tail -3 gpurun_out/r03b/stress.log
for sh in "32 32 32 32 3 1 1 res" "32 16 64 64 3 1 1 res" "8 64 32 32 3 1 1 res" "8 32 64 64 3 1 1 res"; do
  echo "== $sh" >> gpurun_out/r03b/ab.log
  PCC_BENCH_IMPL=0 python tools/bench_one.py $sh 2>&1 | grep impl >> gpurun_out/r03b/ab.log
  PCC_WINO_PER_GROUP=1 PCC_BENCH_IMPL=0 python tools/bench_one.py $sh 2>&1 | grep impl | sed 's/^/per-group /' >> gpurun_out/r03b/ab.log
  PCC_WINO_MULTI=1 PCC_BENCH_IMPL=0 python tools/bench_one.py $sh 2>&1 | grep impl | sed 's/^/old-multi /' >> gpurun_out/r03b/ab.log
done
cat gpurun_out/r03b/ab.log
