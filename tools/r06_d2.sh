R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out/r06d2
timeout 1200 python -m pytest tests/test_round6_gpu.py -x -q -m gpu -k "pruned" -s 2>&1 | tail -4
timeout 1500 python tools/d2_tie_table.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06d2/d2_tie_table.md | tail -12
timeout 1500 python tools/bench_search_cloud.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06d2/bench_search_cloud.log | tail -4
