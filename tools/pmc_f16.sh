R=$PWD; OUT=$R/gpurun_out/r02g; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
CMD="python $R/tools/bench_f16.py 16 32 64 res"
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU --output-format csv -d $OUT/sq -o p -- timeout 180 $CMD > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -o p -- timeout 180 $CMD > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/write -o p -- timeout 180 $CMD > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS --output-format csv -d $OUT/mix -o p -- timeout 180 $CMD > /dev/null 2>&1
python - <<PY
import csv, glob, collections
for d in ['sq','fetch','write','mix']:
    for f in glob.glob('$OUT/'+d+'/**/*counter_collection.csv', recursive=True):
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if 'conv_f16' in r['Kernel_Name']: acc[r['Counter_Name']].append(float(r['Counter_Value']))
        print(d, {k: round(sum(v)/len(v)) for k, v in acc.items()})
PY
