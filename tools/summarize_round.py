"""Turns gpurun_out/<tag>/ (tools/profile_round.sh) into the committed summaries under profiles/.

    python tools/summarize_round.py <tag>                 everything
    python tools/summarize_round.py <tag> --traffic-only  only profiles/dominant_kernel_traffic.json (run ON the GPU box by profile_round.sh right
                                                          after the counter passes of the dominant kernel, so that the bench runs that follow embed
                                                          a traffic_profiled object that is not stale)"""
import collections, csv, glob, json, os, re, shutil, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import summarize_trace as ST
tag = sys.argv[1] if len(sys.argv) > 1 else 'r05'
TRAFFIC_ONLY = '--traffic-only' in sys.argv
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src, dst = os.path.join(root, 'gpurun_out', tag), os.path.join(root, 'profiles')
os.makedirs(dst, exist_ok=True)
TRACE_STEPS = int(open(os.path.join(src, 'trace_steps.txt')).read()) if os.path.exists(os.path.join(src, 'trace_steps.txt')) else 6
if not TRAFFIC_ONLY:
    shutil.copy(glob.glob(os.path.join(src, 'trace', '**', 't_kernel_stats.csv'), recursive=True)[0], os.path.join(dst, f'{tag}_bench_kernel_stats.csv'))
    trace_agg = ST.load(glob.glob(os.path.join(src, 'trace', '**', 't_kernel_trace.csv'), recursive=True)[0])
    summary, trace_stats = ST.table(trace_agg, TRACE_STEPS, 34)
    bench_prof = [l for l in open(os.path.join(src, 'bench_profiled.log')) if l.startswith('{"metric"')]
    bench = [l for l in open(os.path.join(src, 'bench.log')) if l.startswith('{"metric"')]
PEAK_TF, HBM_TBS = 157.3, 8.0
B = 32
# key -> (kernel name fragment, description, algorithmic flops per launch, executed-MFMA flops per launch (None = counter), algorithmic bytes, bound)
KERNELS = {
    'wino16': ('conv16_wino_f16s_kernel<true, false, 1, false>', 'Conv3DTranspose 16->16 k3 s1 @64^3 + residual, batch 32 (two-piece fp16 Winograd F(2x2,3x3) x-y + direct z, conv_wino_f16s.hip)',
               2.0 * B * 64 ** 3 * 27 * 16 * 16, B * 64 ** 3 * 16 * 4 * 3, 'hbm'),
    'wino16_fp32': ('conv16_wino_kernel', 'the same layer on the exact-fp32 MFMA Winograd kernel (PCC_NO_SPLIT=1, conv_wino.hip): the A/B line of the split path',
                    2.0 * B * 64 ** 3 * 27 * 16 * 16, B * 64 ** 3 * 16 * 4 * 3, 'mfma'),
    'wino16_bf16': ('conv16_wino_bf16_kernel', 'the same layer on the three-piece bf16 kernel of round 4 (PCC_NO_F16S=1, conv_wino_bf16.hip)',
                    2.0 * B * 64 ** 3 * 27 * 16 * 16, B * 64 ** 3 * 16 * 4 * 3, 'hbm'),
    'cin32': ('conv16_wino_f16s_kernel<true, false, 2, false>', 'Conv3DTranspose 32->32 k3 s1 @32^3 + residual, batch 32 (two-piece fp16 Winograd, cin groups inside the z march; round 5: conv16_wino_cin_kernel<2>, exact fp32)',
              2.0 * B * 32 ** 3 * 27 * 32 * 32, B * 32 ** 3 * 32 * 4 * 3, 'mfma'),
    'cin64': ('conv16_wino_f16s_kernel<true, false, 2, true>', 'Conv3DTranspose 64->64 k3 s1 @16^3 + residual, batch 32: the SECOND of its two launches (cin groups 2, 3 + the partial sums of the first; two-piece fp16 Winograd; round 5: conv_k3s1_split_kernel<64>, one launch of 117-124 us)',
              2.0 * B * 16 ** 3 * 27 * 64 * 64 / 2, B * 16 ** 3 * 64 * 4 * 3, 'mfma'),
    'cin64_first': ('conv16_wino_f16s_kernel<false, false, 2, false>', 'the FIRST launch of the same layer (cin groups 0, 1 -> raw partial sums)',
              2.0 * B * 16 ** 3 * 27 * 64 * 64 / 2, B * 16 ** 3 * 64 * 4 * 2, 'mfma'),
    'tr2m': ('conv_tr2m_f16s_kernel<2', 'Conv3DTranspose 32->16 k3 s2 32^3 -> 64^3, batch 32 (parity-decomposed, z-marching, two-piece fp16 operands, LDS-resident weights; round 5: conv_tr2m_bf16_kernel)',
             2.0 * B * 32 ** 3 * 27 * 32 * 16, B * (32 ** 3 * 32 + 64 ** 3 * 16) * 4, 'mfma'),
    'tr2g': ('conv_tr2m_f16s_kernel<4', 'Conv3DTranspose 64->32 k3 s2 16^3 -> 32^3, batch 32 (z-marching, two-piece fp16 operands, 110 KB of weight pieces LDS-resident; round 5: conv_tr2_split_kernel, 94 us)',
             2.0 * B * 16 ** 3 * 27 * 64 * 32, B * (16 ** 3 * 64 + 32 ** 3 * 32) * 4, 'mfma'),
    'fwd64_8': ('conv_k3s1_split_kernel<64, 1', 'Conv3D 64->64 k3 s1 @8^3, batch 32 (direct, split-bf16 operands, 8-wide rows; round 3: conv_fwd_kernel)',
                2.0 * B * 8 ** 3 * 27 * 64 * 64, B * 8 ** 3 * 64 * 4 * 2, 'mfma'),
    'cout1': ('conv_cout1_mfma_kernel', 'Conv3DTranspose 16->1 k3 s1 @64^3, batch 32 (tap-plane MFMA + LDS gather, 32 x 32 columns)',
              2.0 * B * 64 ** 3 * 27 * 16, B * 64 ** 3 * (16 + 1) * 4, 'hbm'),
}
rows, traffic_json = [], None
for key, (frag, desc, alg_flops, alg_bytes, bound) in KERNELS.items():
    if TRAFFIC_ONLY and key != 'wino16':
        continue
    if not os.path.exists(os.path.join(src, key, 'time.log')):
        continue
    pmc, dur = {}, []
    for d in ('fetch', 'write', 'sq', 'lds'):
        for f in glob.glob(os.path.join(src, key, d, '**', '*counter_collection.csv'), recursive=True):
            agg = collections.defaultdict(list)
            for r in csv.DictReader(open(f)):
                if frag in r['Kernel_Name']:
                    agg[r['Counter_Name']].append(float(r['Counter_Value']))
            for k, v in agg.items():
                pmc[k] = sum(v) / len(v)
        if d == 'sq':   # launch duration of the profiled (counter) pass, from its own kernel trace
            for f in glob.glob(os.path.join(src, key, d, '**', '*kernel_trace.csv'), recursive=True):
                for r in csv.DictReader(open(f)):
                    if frag in r['Kernel_Name']:
                        dur.append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
    if not dur:
        print('no launch of', frag, 'in the counter pass of', key, file=sys.stderr)
        continue
    tl = open(os.path.join(src, key, 'time.log')).read()
    m = re.search(r'min ([\d.]+) us median ([\d.]+) us', tl)
    t_min, t_med = (float(m.group(1)), float(m.group(2))) if m else (float('nan'),) * 2
    t_call_min, t_call_med = t_min, t_med
    # round 6: a stand-alone pcc_conv3d call of a two-piece fp16 layer runs pcc_block_amax in front of the kernel (inside pcc_network_forward the
    # producer records the maxima), and the 64-channel layer is two launches: bench_one's HIP events time the CALL.  The kernel's own
    # duration comes from the kernel trace of the counter pass instead (profiled clock: a few % slower than un-profiled)
    if key in ('wino16', 'cin32', 'cin64', 'cin64_first', 'tr2m', 'tr2g') and dur:
        sd = sorted(dur)
        t_min, t_med = sd[0], sd[len(sd) // 2]
    fetch = pmc.get('FETCH_SIZE', float('nan')) * 1024 * 2      # KB -> B, x2: gfx950 FETCH_SIZE counts 64 B per 128 B request (MI355X_MICROARCH.md)
    write = pmc.get('WRITE_SIZE', float('nan')) * 1024
    exec_flops = pmc.get('SQ_INSTS_VALU_MFMA_MOPS_F32', float('nan')) * 512
    bf16 = key in ('wino16', 'wino16_bf16', 'cin32', 'cin64', 'cin64_first', 'tr2m', 'tr2g', 'fwd64_8')
    if bf16:     # bf16 MFMAs: v_mfma_f32_16x16x32_bf16 = 16384 flops each (SQ_INSTS_MFMA counts instructions per wave)
        exec_flops = pmc.get('SQ_INSTS_MFMA', float('nan')) * 16384
    simd_cycles = pmc.get('GRBM_GUI_ACTIVE', float('nan')) / 8 * 1024
    out = {'key': key, 'kernel': frag, 'layer': desc, 'bound': bound,
           'launch_us_unprofiled_min': t_min, 'launch_us_unprofiled_median': t_med,
           'call_us_bench_one_min_median': [t_call_min, t_call_med],
           'launch_us_in_counter_pass': sum(dur) / max(len(dur), 1),
           'algorithmic_flops_per_launch': alg_flops, 'executed_mfma_flops_per_launch': exec_flops,
           'mfma_pipe': ('f16 (two-piece operands: 4x the multiply-adds of the fp32 kernel)' if key in ('wino16', 'cin32', 'cin64', 'cin64_first', 'tr2m', 'tr2g') else 'bf16 (split operands: 6x the multiply-adds of the fp32 kernel)') if bf16 else 'fp32',
           'executed_tflops': exec_flops / (t_med * 1e-6) / 1e12, 'executed_frac_of_pipe_peak': exec_flops / (t_med * 1e-6) / 1e12 / (2500.0 if bf16 else PEAK_TF),
           'algorithmic_tflops': alg_flops / (t_med * 1e-6) / 1e12,
           'mfma_busy_frac_of_simd_cycles': pmc.get('SQ_VALU_MFMA_BUSY_CYCLES', float('nan')) / simd_cycles,
           'mfma_busy_frac_of_wave_cycles': pmc.get('SQ_VALU_MFMA_BUSY_CYCLES', float('nan')) / 4 / pmc.get('SQ_WAVE_CYCLES', float('nan')),
           'shader_clock_ghz_in_counter_pass': pmc.get('GRBM_GUI_ACTIVE', float('nan')) / 8 / (sum(dur) / max(len(dur), 1)) / 1e3,
           'algorithmic_bytes_per_launch': alg_bytes, 'hbm_bytes_per_launch': fetch + write, 'fetch_bytes_corrected_x2': fetch, 'write_bytes': write,
           'traffic_over_algorithmic': (fetch + write) / alg_bytes,
           'hbm_tbs_algorithmic': alg_bytes / (t_med * 1e-6) / 1e12, 'hbm_frac_of_8tbs': alg_bytes / (t_med * 1e-6) / 1e12 / HBM_TBS,
           'l2_hit_rate': pmc.get('TCC_HIT_sum', 0) / max(pmc.get('TCC_HIT_sum', 0) + pmc.get('TCC_MISS_sum', 0), 1),
           'wait_any_frac': pmc.get('SQ_WAIT_ANY', float('nan')) / pmc.get('SQ_WAVE_CYCLES', float('nan')),
           'lds_bank_conflict_frac_of_lds_cycles': pmc.get('SQ_LDS_BANK_CONFLICT', 0) / max(pmc.get('SQ_LDS_IDX_ACTIVE', 1), 1),
           'raw_counters': pmc}
    rows.append(out)
    if key == 'wino16':
        import hashlib
        ksrc = 'conv_wino_f16s.hip'
        data = open(os.path.join(root, 'pcc_geo_cnn_v2_amd', 'csrc', ksrc), 'rb').read()
        out['executed_frac_of_bf16_mfma_peak'] = out['executed_tflops'] / 2500.0
        traffic_json = dict(out, kernel_source=ksrc, kernel_source_sha1=hashlib.sha1(b'blob %d\0' % len(data) + data).hexdigest(), method='rocprofv3 --pmc, one pass per counter group, on tools/bench_one.py 32 64 16 16 3 1 1 res; FETCH_SIZE doubled per '
                            'MI355X_MICROARCH.md (gfx950 counts 64 B per 128 B fabric request); WRITE_SIZE as reported (uncalibrated)')
json.dump(traffic_json, open(os.path.join(dst, 'dominant_kernel_traffic.json'), 'w'), indent=1)
if TRAFFIC_ONLY:
    sys.exit(0)
json.dump(rows, open(os.path.join(dst, f'{tag}_kernel_counters.json'), 'w'), indent=1)
# the dominant kernel inside the traced bench (its 64^3 launches = the (kernel, grid) row with the largest total among its rows)
dom_rows = [(k, v) for k, v in trace_stats.items() if 'conv16_wino_f16s_kernel<true, false, 1, false>' in k[0] or 'conv16_wino_bf16' in k[0]]
dom_key, dom = max(dom_rows, key=lambda kv: kv[1]['total']) if dom_rows else (None, None)
DOM_BYTES = 3.0 * B * 64 ** 3 * 16 * 4
DOM_EXEC_BF16 = 4.0 * 2.0 * B * 64 ** 3 * 27 * 16 * 16 * (16.0 / 36.0) * ((64 + 1 - 4.0 / 3.0) / 64)      # bench.py wino_exec_factor(16, 64, 32) x four fp16 product terms (two K = 32 MFMAs per row)
trace_cmd = open(os.path.join(src, 'trace_cmd.txt')).read().strip() if os.path.exists(os.path.join(src, 'trace_cmd.txt')) else 'python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-secondary'
with open(os.path.join(dst, f'{tag}_bench_kernel_summary.md'), 'w') as f:
    f.write(f'# {tag}: `rocprofv3 --kernel-trace --stats -- {trace_cmd}`\n\n')
    f.write(f'{TRACE_STEPS} steps of 32 blocks in the traced process (set-up priming + warm-up + timed, all counted), c3p @64^3, per (kernel, grid size).  '
            'avg = what `--stats` prints (includes the first launch of every kernel, which pays the code-object load); median / trimmed mean = the steady state:\n\n' + summary + '\n\n')
    if dom is not None:
        med, trim = dom['median'] * 1e-9, dom['trimmed'] * 1e-9
        f.write('## The roofline numbers of the bench line, recomputed from THIS trace\n\n')
        f.write(f'Dominant kernel `{dom_key[0]}` (grid {dom_key[1]}): {dom["calls"]} launches = {dom["calls"] / TRACE_STEPS:.2f} per step, median {med*1e6:.1f} us, trimmed mean {trim*1e6:.1f} us, '
                f'average {dom["avg"]/1e3:.1f} us, max {dom["max"]/1e3:.1f} us.\n\n')
        f.write('| from | launch us | algorithmic bytes / launch | TB/s | `roofline.frac` (of 8 TB/s) | executed bf16 GFLOP / launch | TFLOP/s | frac of 2.5 PF (`roofline.mfma.frac_of_bf16_mfma_peak`) |\n|---|---|---|---|---|---|---|---|\n')
        for name, t in (('trace median', med), ('trace trimmed mean', trim)):
            f.write(f'| {name} | {t*1e6:.1f} | {DOM_BYTES/1e9:.4f} GB | {DOM_BYTES/t/1e12:.2f} | **{DOM_BYTES/t/8e12:.3f}** | {DOM_EXEC_BF16/1e9:.1f} | {DOM_EXEC_BF16/t/1e12:.0f} | {DOM_EXEC_BF16/t/2.5e15:.3f} |\n')
        for name, lines in (('bench.py under the profiler (HIP events)', bench_prof), ('bench.py without the profiler (HIP events)', bench)):
            if lines:
                r = json.loads(lines[-1])['roofline']
                t = r['avg_launch_ms'] * 1e-3
                f.write(f'| {name} | {t*1e6:.1f} | {r["algorithmic_bytes_per_launch"]/1e9:.4f} GB | {r["algorithmic_bytes_per_launch"]/t/1e12:.2f} | **{r["frac"]:.3f}** | '
                        f'{r.get("mfma", {}).get("executed_bf16_flops_per_launch", float("nan"))/1e9:.1f} | {r.get("mfma", {}).get("executed_bf16_tflops", float("nan")):.0f} | {r.get("mfma", {}).get("frac_of_bf16_mfma_peak", float("nan")):.3f} |\n')
        f.write('\n')
    f.write('bench.py JSON under the profiler:\n\n```\n' + ''.join(bench_prof) + '```\n\nbench.py JSON without the profiler (same box):\n\n```\n' + ''.join(bench) + '```\n\n')
    f.write('## Counters of the kernels the verdicts name (separate `--pmc` passes on `tools/bench_one.py`, batch 32)\n\n')
    f.write('| kernel | launch us (median, un-profiled) | executed MFMA GFLOP | executed frac of the peak of its pipe (157.3 TF fp32 or 2500 TF bf16) | MFMA busy / SIMD cycles | MFMA busy / wave cycles | clock GHz (counter pass) | '
            'HBM bytes / algorithmic | algorithmic TB/s (frac of 8) | wait_any | LDS conflict frac |\n|---|---|---|---|---|---|---|---|---|---|---|\n')
    for o in rows:
        f.write(f"| `{o['kernel']}` {o['layer']} | {o['launch_us_unprofiled_median']:.1f} | {o['executed_mfma_flops_per_launch']/1e9:.2f} | {o['executed_frac_of_pipe_peak']:.3f} | "
                f"{o['mfma_busy_frac_of_simd_cycles']:.3f} | {o['mfma_busy_frac_of_wave_cycles']:.3f} | {o['shader_clock_ghz_in_counter_pass']:.2f} | {o['traffic_over_algorithmic']:.3f} | "
                f"{o['hbm_tbs_algorithmic']:.2f} ({o['hbm_frac_of_8tbs']:.2f}) | {o['wait_any_frac']:.3f} | {o['lds_bank_conflict_frac_of_lds_cycles']:.3f} |\n")
    f.write(f'\nFull counter sets: `profiles/{tag}_kernel_counters.json`.  `executed MFMA GFLOP` = SQ_INSTS_VALU_MFMA_MOPS_F32 x 512 (fp32 kernels) / SQ_INSTS_MFMA x 16384 (split-bf16 kernels: three v_mfma_f32_16x16x32_bf16 per fp32-equivalent product row); HBM bytes = FETCH_SIZE x 2 (gfx950 '
            'correction, MI355X_MICROARCH.md) + WRITE_SIZE.\n')
print(json.dumps([{k: o[k] for k in ('key', 'launch_us_unprofiled_median', 'executed_frac_of_pipe_peak', 'mfma_busy_frac_of_simd_cycles', 'traffic_over_algorithmic', 'hbm_frac_of_8tbs')} for o in rows], indent=1))
