#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 evidence for profiles/.  Usage: tools/profile_round.sh <tag>
set -u
TAG=${1:-r01}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
# (1) per-kernel time of the benchmark command
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- python $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline > $OUT/bench_profiled.log 2>&1
# (2) counters of the dominant kernel, separate passes (no trace domains mixed with --pmc)
DOM="python $R/tools/bench_one.py 32 64 16 16 3 1 1 res"
PCC_BENCH_IMPL=0 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o p -- $DOM > /dev/null 2>&1
PCC_BENCH_IMPL=0 rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/pmc_write -o p -- $DOM > /dev/null 2>&1
PCC_BENCH_IMPL=0 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F32 --output-format csv -d $OUT/pmc_sq -o p -- $DOM > /dev/null 2>&1
PCC_BENCH_IMPL=0 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU --output-format csv -d $OUT/pmc_lds -o p -- $DOM > /dev/null 2>&1
# (3) un-profiled bench line for comparison
cd $R && python bench.py --steps 10 --warmup 2 > $OUT/bench.log 2>&1
tail -1 $OUT/bench.log | cut -c1-400
