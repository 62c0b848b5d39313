# On the GPU box: SQ counters of conv_tr2g_kernel (Conv3DTranspose 32->16 k3 s2 32^3 -> 64^3, batch 32).  Usage: tools/tr2g_pmc.sh [tag]
R=$PWD; export TMPDIR=/tmp PCC_BENCH_IMPL=0; cd /tmp
OUT=$R/gpurun_out/${1:-tr2g}; mkdir -p $OUT
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU --output-format csv -d $OUT/a -o p -- timeout 120 python $R/tools/bench_one.py 32 32 32 16 3 2 1 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_MFMA SQ_INSTS_VMEM --output-format csv -d $OUT/b -o p -- timeout 120 python $R/tools/bench_one.py 32 32 32 16 3 2 1 > /dev/null 2>&1
python - <<PY
import csv, glob, collections
c = {}
for f in glob.glob('$OUT/**/*counter_collection.csv', recursive=True):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if 'tr2g' in r['Kernel_Name']: acc[r['Counter_Name']].append(float(r['Counter_Value']))
    c.update({k: sum(v)/len(v) for k, v in acc.items()})
wc = c['SQ_WAVE_CYCLES']
print('wave_cycles %.1fM grbm %.3fM mfma_busy/simd %.3f wait_any %.3f wait_inst %.3f active %.3f' % (wc/1e6, c['GRBM_GUI_ACTIVE']/8e6, c['SQ_VALU_MFMA_BUSY_CYCLES']/(c['GRBM_GUI_ACTIVE']/8*1024), c['SQ_WAIT_ANY']/wc, c['SQ_WAIT_INST_ANY']/wc, c['SQ_ACTIVE_INST_ANY']/wc))
print('insts per MFMA: valu %.2f salu %.2f lds %.3f vmem %.3f; lds conflict frac %.2f' % tuple([c[k]/c['SQ_INSTS_MFMA'] for k in ('SQ_INSTS_VALU','SQ_INSTS_SALU','SQ_INSTS_LDS','SQ_INSTS_VMEM')] + [c['SQ_LDS_BANK_CONFLICT']/max(c['SQ_LDS_IDX_ACTIVE'],1)]))
PY
timeout 120 python $R/tools/bench_one.py 32 32 32 16 3 2 1 2>&1 | tail -1
