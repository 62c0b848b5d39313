mkdir -p gpurun_out/r03p
timeout 900 python tools/stress_tr2.py 21 120 > gpurun_out/r03p/stress.log 2>&1; tail -4 gpurun_out/r03p/stress.log
python -m pytest tests/test_conv_gpu.py tests/test_network_gpu.py -m gpu -x -q > gpurun_out/r03p/tests.log 2>&1; tail -3 gpurun_out/r03p/tests.log
for sh in "32 64 16 1 3 1 1" "8 128 16 1 3 1 1"; do
  PCC_BENCH_IMPL=0 python tools/bench_one.py $sh 2>&1 | grep impl
  PCC_COUT1_T16=1 PCC_BENCH_IMPL=0 python tools/bench_one.py $sh 2>&1 | grep impl | sed 's/^/t16 /'
done
tools/pmc_shape.sh r03p cout1_t32 "32 64 16 1 3 1 1"
