# the fused linear-time z + y distance transform (k_edt_zy) against the two-kernel form (PCC_EDT_OLD=1): exactness tests, then the adaptive search per cloud
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
timeout 1500 python -m pytest tests/test_threshold_search_gpu.py tests/test_codec_gpu.py tests/test_round6_gpu.py -x -q -m gpu 2>&1 | tail -4
bash tools/r06_search_trace.sh 2>&1 | grep -v "^\"void pcc\|rocclr" | head -12
PCC_EDT_OLD=1 bash tools/r06_search_trace.sh 2>&1 | grep "adaptive"
