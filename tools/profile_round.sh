#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 evidence for profiles/.  Usage: tools/profile_round.sh <tag>
# No trace domain other than --kernel-trace is ever combined with --pmc (gpurun refuses that), counters in separate passes.
set -u
TAG=${1:-r06}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
# (2) counters of the kernels VERDICT names, separate passes each: <key> <bench_one shape>
pmc() {
  KEY=$1; shift
  CMD="python $R/tools/bench_one.py $*"
  PCC_BENCH_IMPL=0 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/$KEY/fetch -o p -- timeout 180 $CMD > /dev/null 2>&1
  PCC_BENCH_IMPL=0 rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/$KEY/write -o p -- timeout 180 $CMD > /dev/null 2>&1
  PCC_BENCH_IMPL=0 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F32 --output-format csv -d $OUT/$KEY/sq -o p -- timeout 180 $CMD > /dev/null 2>&1
  PCC_BENCH_IMPL=0 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_VMEM SQ_INSTS_SALU --output-format csv -d $OUT/$KEY/lds -o p -- timeout 180 $CMD > /dev/null 2>&1
  # un-profiled duration of the same launch (HIP events inside bench_one)
  PCC_BENCH_IMPL=0 timeout 120 $CMD 2>&1 | grep -v amdgpu.ids > $OUT/$KEY/time.log
}
pmc wino16 32 64 16 16 3 1 1 res
# profiles/dominant_kernel_traffic.json from these passes BEFORE any bench run below, so that their traffic_profiled object is not stale
python $R/tools/summarize_round.py $TAG --traffic-only && cp $R/profiles/dominant_kernel_traffic.json $OUT/dominant_kernel_traffic.json
# (1) per-kernel time of the benchmark command: 15 steps, all of them traced and counted (no set-up priming, no A/B, no secondary)
TRACE_CMD="python bench.py --steps 10 --warmup 5 --no-cpu-baseline --no-secondary --no-ab"
echo "PCC_BENCH_NO_STEP_PROFILE=1 PCC_BENCH_NO_PRIME=1 $TRACE_CMD" > $OUT/trace_cmd.txt; echo 15 > $OUT/trace_steps.txt
PCC_BENCH_NO_STEP_PROFILE=1 PCC_BENCH_NO_PRIME=1 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- timeout 180 python $R/bench.py --steps 10 --warmup 5 --no-cpu-baseline --no-secondary --no-ab > $OUT/bench_profiled.log 2>&1
export PCC_NO_SPLIT=1; pmc wino16_fp32 32 64 16 16 3 1 1 res; unset PCC_NO_SPLIT
export PCC_NO_F16S=1; pmc wino16_bf16 32 64 16 16 3 1 1 res; unset PCC_NO_F16S
pmc cin32 32 32 32 32 3 1 1 res
pmc cin64 32 16 64 64 3 1 1 res
mkdir -p $OUT/cin64_first && for d in fetch write sq lds; do cp -r $OUT/cin64/$d $OUT/cin64_first/; done; cp $OUT/cin64/time.log $OUT/cin64_first/time.log
pmc tr2m 32 32 32 16 3 2 1
pmc tr2g 32 16 64 32 3 2 1
pmc cout1 32 64 16 1 3 1 1
pmc fwd64_8 32 8 64 64 3 1 0
# (3) un-profiled bench line for comparison
cd $R && python bench.py --steps 20 --warmup 5 > $OUT/bench.log 2>&1
# keep what travels back small
find $OUT -name "*agent_info.csv" -delete
tail -1 $OUT/bench.log | cut -c1-300
