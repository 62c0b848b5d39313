mkdir -p gpurun_out/r03o
python -m pytest tests/test_conv_gpu.py -m gpu -x -q -k "winograd" > gpurun_out/r03o/tests.log 2>&1; tail -3 gpurun_out/r03o/tests.log
python tools/stress_wino.py 9 80 > gpurun_out/r03o/stress.log 2>&1; tail -2 gpurun_out/r03o/stress.log
bash tools/ab_run.sh r03o_t "32 32 32 32 3 1 1 res;32 16 64 64 3 1 1 res;32 64 16 16 3 1 1 res;32 32 16 16 3 1 0" base dz0 base dz0
