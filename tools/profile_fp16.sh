#!/bin/bash
# GPU box: rocprofv3 evidence of the fp16 mode (BASELINE.json configs[4]) -> gpurun_out/<tag>/; tools/summarize_fp16.py <tag> condenses it.
set -u
TAG=${1:-r05_fp16}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
# 15 steps, all traced and counted (no set-up priming)
PCC_BENCH_NO_PRIME=1 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- timeout 300 python $R/bench.py --workload configs4 --steps 10 --warmup 5 > $OUT/bench_profiled.log 2>&1
CMD="timeout 180 python $R/tools/bench_f16.py 16 8 128 res"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -o p -- $CMD > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/write -o p -- $CMD > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU --output-format csv -d $OUT/sq -o p -- $CMD > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $OUT/mix -o p -- $CMD > /dev/null 2>&1
$CMD 2>&1 | grep -v amdgpu.ids > $OUT/time.log
cd $R && timeout 300 python bench.py --workload configs4 --steps 20 --warmup 3 > $OUT/bench.log 2>&1
find $OUT -name "*agent_info.csv" -delete
tail -1 $OUT/bench.log | cut -c1-200
