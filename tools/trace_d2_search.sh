export TMPDIR=/tmp; mkdir -p $GRAFT_REPO_ROOT/gpurun_out/r05_d2trace; cd /tmp
PCC_D2_GPU=1 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r05_d2trace -o t -- python $GRAFT_REPO_ROOT/tools/bench_search_cloud.py > $GRAFT_REPO_ROOT/gpurun_out/r05_d2trace/log.txt 2>&1
tail -3 $GRAFT_REPO_ROOT/gpurun_out/r05_d2trace/log.txt
