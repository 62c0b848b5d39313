#!/bin/bash
# GPU box: SQ/LDS counters of one layer shape.  Usage: tools/pmc_one.sh <tag> <impl> <bench_one args...>
TAG=$1; IMPL=$2; shift 2
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp PCC_BENCH_IMPL=$IMPL
cd /tmp
CMD="python $R/tools/bench_one.py $*"
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F32 --output-format csv -d $OUT/pmc_sq -o p -- timeout 180 $CMD > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM --output-format csv -d $OUT/pmc_lds -o p -- timeout 180 $CMD > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INSTS_SALU SQ_INSTS_MFMA SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES --output-format csv -d $OUT/pmc_misc -o p -- timeout 180 $CMD > /dev/null 2>&1
python - <<PY
import csv, glob, collections
for d in ['pmc_sq', 'pmc_lds', 'pmc_misc']:
    for f in glob.glob('$OUT/' + d + '/**/*counter_collection.csv', recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
        for r in csv.DictReader(open(f)):
            k = r['Kernel_Name'][:60]
            acc[k][r['Counter_Name']] += float(r['Counter_Value']); n[(k, r['Counter_Name'])] += 1
        for k in acc:
            if 'conv' not in k: continue
            print(d, k, {c: round(v / n[(k, c)]) for c, v in acc[k].items()})
PY
