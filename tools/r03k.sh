mkdir -p gpurun_out/r03k
timeout 600 python tools/stress_tr2.py 3 150 > gpurun_out/r03k/stress.log 2>&1; tail -6 gpurun_out/r03k/stress.log
for sh in "32 32 32 16 3 2 1" "32 16 64 32 3 2 1" "8 64 32 16 3 2 1"; do
  echo "== $sh" >> gpurun_out/r03k/ab.log
  PCC_BENCH_IMPL=0 timeout 120 python tools/bench_one.py $sh 2>&1 | grep impl >> gpurun_out/r03k/ab.log
  PCC_NO_TR2M=1 PCC_BENCH_IMPL=0 timeout 120 python tools/bench_one.py $sh 2>&1 | grep impl | sed 's/^/tr2g /' >> gpurun_out/r03k/ab.log
done
cat gpurun_out/r03k/ab.log
tools/pmc_shape.sh r03k tr2m_32_16 "32 32 32 16 3 2 1"
tools/pmc_shape.sh r03k tr2m_64_32 "32 16 64 32 3 2 1"
