"""Turns gpurun_out/<tag>/ (tools/profile_round.sh) into the committed summaries under profiles/."""
import collections, csv, glob, json, os, shutil, subprocess, sys
tag = sys.argv[1] if len(sys.argv) > 1 else 'r01'
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src, dst = os.path.join(root, 'gpurun_out', tag), os.path.join(root, 'profiles')
os.makedirs(dst, exist_ok=True)
shutil.copy(os.path.join(src, 'trace', 't_kernel_stats.csv'), os.path.join(dst, f'{tag}_bench_kernel_stats.csv'))
summary = subprocess.check_output([sys.executable, os.path.join(root, 'tools', 'summarize_trace.py'),
                                   os.path.join(src, 'trace', 't_kernel_trace.csv'), '6', '34'], text=True)
bench_prof = [l for l in open(os.path.join(src, 'bench_profiled.log')) if l.startswith('{"metric"')]
bench = [l for l in open(os.path.join(src, 'bench.log')) if l.startswith('{"metric"')]
pmc = {}
for d in ('pmc_fetch', 'pmc_write', 'pmc_sq', 'pmc_lds'):
    for f in glob.glob(os.path.join(src, d, '*counter_collection.csv')):
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if 'conv16_wino' in r['Kernel_Name'] or 'conv16_pers' in r['Kernel_Name']:
                agg[r['Counter_Name']].append(float(r['Counter_Value']))
        for k, v in agg.items():
            pmc[k] = sum(v) / len(v)
B = 32
alg = B * 64 ** 3 * 16 * 4 * 3            # in + residual + out, bytes per launch
fetch = pmc.get('FETCH_SIZE', float('nan')) * 1024 * 2   # KB -> B, x2: gfx950 FETCH_SIZE counts 64 B per 128 B request
write = pmc.get('WRITE_SIZE', float('nan')) * 1024
traffic = fetch + write
simd_cycles = pmc.get('GRBM_GUI_ACTIVE', float('nan')) / 8 * 1024   # per-XCD cycles x 1024 SIMDs
out = {'kernel': 'conv16_wino_kernel<relu,clip> (Winograd F(2x2,3x3)+direct z), Conv3DTranspose 16->16 k3 s1 @64^3 + residual, batch 32',
       'hbm_bytes_per_launch': traffic, 'fetch_bytes_corrected_x2': fetch, 'write_bytes': write,
       'algorithmic_bytes_per_launch': alg, 'traffic_over_algorithmic': traffic / alg,
       'l2_hit_rate': pmc.get('TCC_HIT_sum', 0) / max(pmc.get('TCC_HIT_sum', 0) + pmc.get('TCC_MISS_sum', 0), 1),
       'mfma_busy_frac': pmc.get('SQ_VALU_MFMA_BUSY_CYCLES', float('nan')) / simd_cycles,
       'lds_bank_conflict_frac_of_lds_cycles': pmc.get('SQ_LDS_BANK_CONFLICT', 0) / max(pmc.get('SQ_LDS_IDX_ACTIVE', 1), 1),
       'raw_counters': pmc,
       'method': 'rocprofv3 --pmc, one pass per counter group, on tools/bench_one.py 32 64 16 16 3 1 1 res; FETCH_SIZE doubled per '
                 'MI355X_MICROARCH.md (gfx950 counts 64 B per 128 B fabric request); WRITE_SIZE as reported (uncalibrated)'}
json.dump(out, open(os.path.join(dst, 'dominant_kernel_traffic.json'), 'w'), indent=1)
with open(os.path.join(dst, f'{tag}_bench_kernel_summary.md'), 'w') as f:
    f.write(f'# {tag}: `rocprofv3 --kernel-trace --stats -- python bench.py --steps 5 --warmup 1 --no-cpu-baseline`\n\n')
    f.write('6 steps of 32 blocks (1 warm-up + 5 timed), c3p @64^3, per (kernel, grid size):\n\n' + summary + '\n')
    f.write('bench.py JSON under the profiler:\n\n```\n' + ''.join(bench_prof) + '```\n\nbench.py JSON without the profiler (same box):\n\n```\n' + ''.join(bench) + '```\n\n')
    f.write('Dominant-kernel counters (separate `--pmc` passes):\n\n```\n' + json.dumps(out, indent=1) + '\n```\n')
print(json.dumps({k: out[k] for k in ('hbm_bytes_per_launch', 'algorithmic_bytes_per_launch', 'traffic_over_algorithmic', 'mfma_busy_frac', 'l2_hit_rate')}, indent=1))
