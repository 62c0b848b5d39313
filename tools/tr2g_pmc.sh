R=$PWD; export TMPDIR=/tmp PCC_BENCH_IMPL=0; cd /tmp
for FL in 0 0x70000000 0x40000000 0x30000000; do
  OUT=$R/gpurun_out/r02m/f$FL; mkdir -p $OUT
  PCC_BENCH_FLAGS=$FL rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU --output-format csv -d $OUT -o p -- timeout 120 python $R/tools/bench_one.py 32 32 32 16 3 2 1 > /dev/null 2>&1
  python - <<PY
import csv, glob, collections
for f in glob.glob('$OUT/**/*counter_collection.csv', recursive=True):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if 'tr2g' in r['Kernel_Name']: acc[r['Counter_Name']].append(float(r['Counter_Value']))
    c = {k: sum(v)/len(v) for k, v in acc.items()}
    wc = c['SQ_WAVE_CYCLES']
    print('flags $FL: wave_cycles %.1fM grbm %.3fM mfma_busy/simd %.3f wait_any %.3f wait_inst %.3f active %.3f valu %.1fM' % (wc/1e6, c['GRBM_GUI_ACTIVE']/8e6, c['SQ_VALU_MFMA_BUSY_CYCLES']/(c['GRBM_GUI_ACTIVE']/8*1024), c['SQ_WAIT_ANY']/wc, c['SQ_WAIT_INST_ANY']/wc, c['SQ_ACTIVE_INST_ANY']/wc, c['SQ_INSTS_VALU']/1e6))
PY
done
