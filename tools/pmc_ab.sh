#!/bin/bash
# GPU box: cycle-level A/B of library variants (wall time moves with the power-limited clock; SQ_WAVE_CYCLES does not).
# usage: tools/pmc_ab.sh <outdir> "<bench_one shape>" <variant> ...     (variants: build_ab/lib<variant>.so)
OUTN=$1; SHAPE=$2; shift 2
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$OUTN
mkdir -p $OUT
export TMPDIR=/tmp PCC_BENCH_IMPL=0
cd /tmp
for v in "$@"; do
  export PCC_GEO_LIB=$R/build_ab/lib$v.so
  rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU --output-format csv -d $OUT/$v -o p -- timeout 180 python $R/tools/bench_one.py $SHAPE > /dev/null 2>&1
  python - <<PY
import csv, glob, collections
for f in glob.glob('$OUT/$v/**/*counter_collection.csv', recursive=True):
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'][:50]
        acc[k][r['Counter_Name']] += float(r['Counter_Value']); n[(k, r['Counter_Name'])] += 1
    for k in acc:
        if 'conv' not in k: continue
        c = {m: v / n[(k, m)] for m, v in acc[k].items()}
        wc = c['SQ_WAVE_CYCLES']
        print(f"$v {k}: wave_cycles {wc/1e6:.2f}M  mfma_busy/wave_cycles {c['SQ_VALU_MFMA_BUSY_CYCLES']/4/wc:.3f}  wait_any {c['SQ_WAIT_ANY']/wc:.3f}  wait_inst {c['SQ_WAIT_INST_ANY']/wc:.3f}  valu_insts {c['SQ_INSTS_VALU']/1e6:.1f}M  grbm {c['GRBM_GUI_ACTIVE']/8e6:.3f}M")
PY
done
