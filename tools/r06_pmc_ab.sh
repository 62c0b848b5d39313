# counters of the 16->16 @64^3 launch for several library builds (tools/pmc_shape.sh per build)
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
for v in "$@"; do
  if [ $v = tree ]; then bash tools/pmc_shape.sh r06pmc $v "32 64 16 16 3 1 1 res" 2>&1 | tail -2
  else bash tools/pmc_shape.sh r06pmc $v "32 64 16 16 3 1 1 res" PCC_GEO_LIB=$R/build_ab/lib$v.so 2>&1 | tail -2; fi
done
rm -rf $R/gpurun_out/r06pmc/*/a $R/gpurun_out/r06pmc/*/b
