mkdir -p gpurun_out/r03d
python -m pytest tests/test_conv_gpu.py -m gpu -x -q -k "winograd" > gpurun_out/r03d/tests.log 2>&1; tail -3 gpurun_out/r03d/tests.log
python tools/stress_wino.py 11 60 > gpurun_out/r03d/stress.log 2>&1; tail -2 gpurun_out/r03d/stress.log
for sh in "32 32 32 32 3 1 1 res" "32 16 64 64 3 1 1 res"; do
  echo "== $sh" >> gpurun_out/r03d/ab.log
  PCC_BENCH_IMPL=0 python tools/bench_one.py $sh 2>&1 | grep impl >> gpurun_out/r03d/ab.log
  PCC_WINO_PER_GROUP=1 PCC_BENCH_IMPL=0 python tools/bench_one.py $sh 2>&1 | grep impl | sed 's/^/per-group /' >> gpurun_out/r03d/ab.log
done
cat gpurun_out/r03d/ab.log
tools/pmc_shape.sh r03d cin32 "32 32 32 32 3 1 1 res"
tools/pmc_shape.sh r03d cin64 "32 16 64 64 3 1 1 res"
