# counter passes of the dominant kernel only -> profiles/dominant_kernel_traffic.json (the subset of tools/profile_round.sh that bench.py's
# traffic_profiled object and tests/test_round4_cpu.py depend on), then a bench line with the step profile
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=r06t; OUT=$R/gpurun_out/$TAG; rm -rf $OUT; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
KEY=wino16; CMD="python $R/tools/bench_one.py 32 64 16 16 3 1 1 res"
PCC_BENCH_IMPL=0 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/$KEY/fetch -o p -- timeout 180 $CMD > /dev/null 2>&1
PCC_BENCH_IMPL=0 rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/$KEY/write -o p -- timeout 180 $CMD > /dev/null 2>&1
PCC_BENCH_IMPL=0 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F32 --output-format csv -d $OUT/$KEY/sq -o p -- timeout 180 $CMD > /dev/null 2>&1
PCC_BENCH_IMPL=0 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_VMEM SQ_INSTS_SALU --output-format csv -d $OUT/$KEY/lds -o p -- timeout 180 $CMD > /dev/null 2>&1
PCC_BENCH_IMPL=0 timeout 120 $CMD 2>&1 | grep -v amdgpu.ids > $OUT/$KEY/time.log
python $R/tools/summarize_round.py $TAG --traffic-only && cp $R/profiles/dominant_kernel_traffic.json $OUT/dominant_kernel_traffic.json
find $OUT -name "*agent_info.csv" -delete; find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*counter_collection.csv" -delete
cd $R && python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-secondary --no-ab > $OUT/bench.log 2>&1; tail -1 $OUT/bench.log | cut -c1-200
