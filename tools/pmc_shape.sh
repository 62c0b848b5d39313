#!/bin/bash
# GPU box: SQ counters of one conv shape with the CURRENT library (optionally with env vars): tools/pmc_shape.sh <outdir> <tag> "<bench_one shape>" [VAR=val ...]
OUTN=$1; TAG=$2; SHAPE=$3; shift 3
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$OUTN/$TAG
mkdir -p $OUT
export TMPDIR=/tmp PCC_BENCH_IMPL=${PCC_BENCH_IMPL:-0}
for kv in "$@"; do export "$kv"; done
cd /tmp
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU --output-format csv -d $OUT/a -o p -- timeout 180 python $R/tools/bench_one.py $SHAPE > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_MFMA --output-format csv -d $OUT/b -o p -- timeout 180 python $R/tools/bench_one.py $SHAPE > /dev/null 2>&1
python - <<PY
import csv, glob, collections
c = collections.defaultdict(dict)
for sub in 'ab':
    for f in glob.glob('$OUT/' + sub + '/**/*counter_collection.csv', recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
        for r in csv.DictReader(open(f)):
            k = r['Kernel_Name'][:60]
            acc[k][r['Counter_Name']] += float(r['Counter_Value']); n[(k, r['Counter_Name'])] += 1
        for k in acc:
            if 'conv' in k:
                for m, v in acc[k].items(): c[k][m] = v / n[(k, m)]
for k, d in c.items():
    wc = d['SQ_WAVE_CYCLES']
    print(f"$TAG {k}\n   wave_cycles {wc/1e6:.2f}M  mfma_busy/wc {d['SQ_VALU_MFMA_BUSY_CYCLES']/4/wc:.3f}  wait_any {d['SQ_WAIT_ANY']/wc:.3f}  wait_inst {d['SQ_WAIT_INST_ANY']/wc:.3f}  active {d['SQ_ACTIVE_INST_ANY']/wc:.3f}  valu {d['SQ_INSTS_VALU']/1e6:.2f}M  mfma {d.get('SQ_INSTS_MFMA',0)/1e6:.2f}M  lds {d.get('SQ_INSTS_LDS',0)/1e6:.2f}M  salu {d.get('SQ_INSTS_SALU',0)/1e6:.2f}M  vmem {d.get('SQ_INSTS_VMEM',0)/1e6:.2f}M  lds_conf {d.get('SQ_LDS_BANK_CONFLICT',0)/max(d.get('SQ_LDS_IDX_ACTIVE',1),1):.3f}  wait_lds {d.get('SQ_WAIT_INST_LDS',0)/wc:.3f}  grbm {d['GRBM_GUI_ACTIVE']/1e6:.3f}M")
PY
