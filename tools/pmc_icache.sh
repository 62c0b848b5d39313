#!/bin/bash
# GPU box: instruction-cache counters of one conv shape: tools/pmc_icache.sh "<bench_one shape>" [VAR=val ...]
SHAPE=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/icache; rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp PCC_BENCH_IMPL=${PCC_BENCH_IMPL:-0}
for kv in "$@"; do export "$kv"; done
cd /tmp
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH SQ_WAIT_INST_ANY SQ_INSTS_SALU --output-format csv -d $OUT/a -o p -- timeout 180 python $R/tools/bench_one.py $SHAPE > $OUT/log.txt 2>&1
python - <<PY
import csv, glob, collections
for f in glob.glob('$OUT/a/**/*counter_collection.csv', recursive=True):
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'][:70]
        acc[k][r['Counter_Name']] += float(r['Counter_Value']); n[(k, r['Counter_Name'])] += 1
    for k in acc:
        if 'conv' in k:
            print(k, {m: round(v / n[(k, m)]) for m, v in acc[k].items()})
PY
tail -2 $OUT/log.txt
