"""gpurun_out/<tag>/ (tools/profile_fp16.sh) -> profiles/<tag>_summary.md + profiles/<tag>_kernel_stats.csv"""
import collections, csv, glob, json, os, re, shutil, subprocess, sys
tag = sys.argv[1] if len(sys.argv) > 1 else 'r02_fp16'
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src, dst = os.path.join(root, 'gpurun_out', tag), os.path.join(root, 'profiles')
shutil.copy(glob.glob(os.path.join(src, 'trace', '**', 't_kernel_stats.csv'), recursive=True)[0], os.path.join(dst, f'{tag}_kernel_stats.csv'))
summary = subprocess.check_output([sys.executable, os.path.join(root, 'tools', 'summarize_trace.py'),
                                   glob.glob(os.path.join(src, 'trace', '**', 't_kernel_trace.csv'), recursive=True)[0], '15', '24'], text=True)
bench = [l for l in open(os.path.join(src, 'bench.log')) if l.startswith('{"metric"')]
bench_prof = [l for l in open(os.path.join(src, 'bench_profiled.log')) if l.startswith('{"metric"')]
pmc, dur = {}, []
for d in ('fetch', 'write', 'sq', 'mix'):
    for f in glob.glob(os.path.join(src, d, '**', '*counter_collection.csv'), recursive=True):
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if 'conv_f16' in r['Kernel_Name']:
                agg[r['Counter_Name']].append(float(r['Counter_Value']))
        for k, v in agg.items():
            pmc[k] = sum(v) / len(v)
    if d == 'sq':
        for f in glob.glob(os.path.join(src, d, '**', '*kernel_trace.csv'), recursive=True):
            dur += [(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3 for r in csv.DictReader(open(f)) if 'conv_f16' in r['Kernel_Name']]
m = re.search(r'min ([\d.]+) us median ([\d.]+) us', open(os.path.join(src, 'time.log')).read())
t_med = float(m.group(2))
N, D, C = 8, 128, 16
alg = 3 * N * D ** 3 * C * 2
fetch, write = pmc['FETCH_SIZE'] * 1024 * 2, pmc['WRITE_SIZE'] * 1024
out = {'kernel': 'conv_f16_kernel<16> (fp16 storage, v_mfma_f32_16x16x32_f16), Conv3DTranspose 16->16 k3 s1 + fp16 residual @128^3, batch 8',
       'bound': 'hbm', 'launch_us_unprofiled_median': t_med, 'launch_us_in_counter_pass': sum(dur) / max(len(dur), 1),
       'algorithmic_bytes_per_launch': alg, 'hbm_bytes_per_launch': fetch + write, 'fetch_bytes_corrected_x2': fetch, 'write_bytes': write,
       'traffic_over_algorithmic': (fetch + write) / alg, 'achieved_tbs_algorithmic': alg / (t_med * 1e-6) / 1e12,
       'frac_of_8tbs': alg / (t_med * 1e-6) / 1e12 / 8.0, 'frac_of_6p3tbs_achievable': alg / (t_med * 1e-6) / 1e12 / 6.3,
       'mfma_busy_frac_of_simd_cycles': pmc['SQ_VALU_MFMA_BUSY_CYCLES'] / (pmc['GRBM_GUI_ACTIVE'] / 8 * 1024),
       'wait_any_frac_of_wave_cycles': pmc['SQ_WAIT_ANY'] / pmc['SQ_WAVE_CYCLES'],
       'lds_bank_conflict_cycles': pmc.get('SQ_LDS_BANK_CONFLICT', 0), 'raw_counters': pmc}
json.dump(out, open(os.path.join(dst, f'{tag}_kernel_counters.json'), 'w'), indent=1)
with open(os.path.join(dst, f'{tag}_summary.md'), 'w') as f:
    f.write(f'# {tag}: fp16 mode, BASELINE.json configs[4] (NOT the headline precision)\n\n`rocprofv3 --kernel-trace --stats -- PCC_BENCH_NO_PRIME=1 python bench.py --workload configs4 --steps 10 --warmup 5` (15 steps, all counted)` '
            '(c3p graph, 128^3 blocks, batch 8, encode+decode), per (kernel, grid size):\n\n' + summary + '\n')
    f.write('`__amd_rocclr_copyBuffer` rows above: the pinned device<->host copies of the side streams (symbols, CDF-row indexes, decoded point lists).  '
            'With rocprofv3 attached the runtime executes them as a blit KERNEL on the CUs; without it they go to the SDMA engines '
            '(`tools/ubench/d2h_copy.hip`), so their share of the kernel time is an artefact of the trace -- the two bench lines below differ by about that share.\n\n')
    f.write('bench.py JSON under the profiler:\n\n```\n' + ''.join(bench_prof) + '```\n\nbench.py JSON without the profiler (same box):\n\n```\n' + ''.join(bench) + '```\n\n')
    f.write('Dominant kernel of the mode, separate `--pmc` passes on `tools/bench_f16.py 16 8 128 res`:\n\n```\n' + json.dumps(out, indent=1) + '\n```\n')
print(json.dumps({k: v for k, v in out.items() if k != 'raw_counters'}, indent=1))
