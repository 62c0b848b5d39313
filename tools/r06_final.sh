# end-of-round check on one fresh box: the whole GPU suite, smoke(), the driver's bench command five times
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out/r06final
timeout 3000 python -m pytest tests/ -x -q -m gpu 2>&1 | tail -4
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
bash tools/r06_driver_cmd.sh 2>&1 | tail -6
cp gpurun_out/r06drv/run3.log gpurun_out/r06final/bench_line.json
