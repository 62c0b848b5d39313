mkdir -p gpurun_out/r03n
python -m pytest tests/test_conv_gpu.py -m gpu -x -q -k "winograd" > gpurun_out/r03n/tests.log 2>&1; tail -3 gpurun_out/r03n/tests.log
python tools/stress_wino.py 5 60 > gpurun_out/r03n/stress.log 2>&1; tail -2 gpurun_out/r03n/stress.log
bash tools/pmc_ab.sh r03n_a "32 32 32 32 3 1 1 res" base mask
bash tools/pmc_ab.sh r03n_b "32 16 64 64 3 1 1 res" base mask
bash tools/ab_run.sh r03n_t "32 32 32 32 3 1 1 res;32 16 64 64 3 1 1 res" base mask
