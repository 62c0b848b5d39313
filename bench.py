#!/usr/bin/env python
"""bench.py -- encode+decode throughput of 64^3 voxel blocks (c3p, fixed threshold) on N MI355X.

One step = one pass of the hot path over one batch of 32 synthetic 64^3 occupancy grids per GPU:
compress graph (analysis, hyper-analysis, quantise, hyper-synthesis, index, quantise, range-encode,
synthesis, threshold+compaction) + decompress graph (range-decode z, hyper-synthesis, index, range-decode y,
synthesis, threshold+compaction, points to host).  Inputs are resident in HBM when timing starts.

  python bench.py --gpus 1 --steps 10 --warmup 2
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \\
         bench.py --gpus N --steps K --warmup W
"""
import argparse
import json
import os
import sys
import time

# The codec keeps four HIP streams busy (kernels; symbols + points to the host; decoder symbols to the device; decoder indexes to
# the host) and RCCL adds its own.  The runtime multiplexes streams onto 4 hardware queues by default: with a process group
# initialised the copy streams then share the kernels' queue and every copy serialises with them (measured at world 1:
# 5378 -> 6104 blocks/s with 8 queues).  Read when the HIP runtime loads, i.e. before `import torch`.
os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')

import numpy as np  # noqa: E402
import torch  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from pcc_geo_cnn_v2_amd import ops  # noqa: E402
from pcc_geo_cnn_v2_amd.model_configs import ModelConfigType  # noqa: E402

RES = 64
BATCH = 32            # blocks per GPU per step (BASELINE.json configs[1]: "c3p, batch=32 random 64^3 grids")
CHUNK = 32            # blocks per pipeline chunk (one chunk per step; steps stream through the pipeline)
PROFILE_STRIDE = 4    # every 4th launch of the dominant layer carries the two HIP events of the live roofline measurement
FLOPS_PER_BLOCK = 31.086e9   # c3p @64^3, compress 16.562 + decompress 14.524 GFLOP
PEAK_FP32_MFMA = 157.3       # TFLOP/s dense, MI355X_MICROARCH.md (v_mfma_f32_16x16x4_f32)
PEAK_BF16_MFMA = 2500.0      # TFLOP/s dense, MI355X_MICROARCH.md (v_mfma_f32_16x16x32_bf16)
PEAK_HBM = 8000.0            # GB/s
# tools/host_budget.sh (DESIGN_HISTORY.md section 6): cores per rank below which the 1-GPU rate drops by more than 3 %
HOST_MIN_CORES_PER_RANK = 3        # late round 4 (4.2 - 4.5 ms steps): 16: 7567 / 7100, 4: 7486 / 7049, 3: 7438 / 7078, 2: 7163 / 6998 blocks/s (profiles/r04_host_budget.log)
# Synthetic weights (no trained checkpoints exist in the container): Glorot-uniform kernels scaled so that the
# coded statistics resemble a trained codec at a high-rate point: ~4 % non-zero y symbols (~1.5-2 KB per
# block), ~5-7 k decoded points per 64^3 block (input: ~5 k points, 2 % occupancy).
GAIN_ANALYSIS, GAIN_SYNTHESIS, FINAL_BIAS, EB_INIT_SCALE = 1.35, 1.8, 0.0, 0.2


def wino_exec_factor(ch, d, batch, num_cu=256):
    """Executed / direct-convolution MFMA flops of a k3 stride-1 layer on csrc/conv_wino.hip: 16 instead of 36 multiplies per 2x2
    outputs and z tap, times the input planes a slab marches.  z-split as in pcc_conv_wino (the split that gives every CU a
    workgroup).  A slab runs (2/3 if it starts at z = 0 else 1) + zlen - (1 if it ends at z = D) plane-equivalents of MFMA rows:
    padding planes are not marched, the two head planes run only the rows that feed this slab; a tail plane inside the volume is a
    full step in the 16-channel kernel and a third of one (its dz = 2 rows) in the multi-group kernels."""
    zs, base = 1, batch * (d // 16) ** 2 * (ch // 16)
    while base * zs < num_cu and d % (zs * 2) == 0 and d // (zs * 2) >= (4 if ch >= 32 else 8):
        zs *= 2
    planes = d + zs - 4.0 / 3.0 if ch == 16 else d + (zs - 2.0) / 3.0
    return 16.0 / 36.0 * planes / d


def c3p_step_flops(res, batch, num_cu=256, winograd=True, split=False):
    """(algorithmic, executed) fp32 MFMA flops of ONE block through compress graph + decompress graph of c3p (the unit of
    SURVEY.md 8d): algorithmic = direct convolution, 2 * MACs as the reference executes them; executed = what the kernels
    issue: the k3 stride-1 layers that csrc/conv_wino.hip takes (conv_mfma.hip dispatch rule: Cin = Cout in {16, 32, 64}, H and W
    multiples of 16) run 16 instead of 36 multiplies per 2x2 outputs and z tap and march
    zlen + 2 input planes per zlen outputs, the first of them with a third of its MFMA rows (zlen = D / z-split, the split that
    gives every CU a workgroup).  split = the round-4 dispatch: the 32- / 64-channel layers on grids <= 16^3 are DIRECT convolutions on
    the bf16 pipe (conv_split.hip: every fp32 product as three bf16 MFMAs), i.e. all 27 taps are executed."""
    def wino_factor(ch, d):
        if not winograd or d % 16 or (split and ch >= 32 and d <= 16):
            return 1.0
        return wino_exec_factor(ch, d, batch, num_cu)
    alg = ex = 0.0

    def conv(cin, cout, k, d_out_or_in, wino=False):       # MACs counted on the grid the taps are applied on
        nonlocal alg, ex
        f = 2.0 * d_out_or_in ** 3 * k ** 3 * cin * cout
        alg += f
        ex += f * (wino_factor(cin, d_out_or_in) if wino else 1.0)

    def analysis():
        d = res
        for f_in, f in ((1, 16), (16, 32), (32, 64)):
            d //= 2
            conv(f_in, f, 3, d)                               # stride-2 conv: taps per output voxel
            conv(f, f, 3, d, True); conv(f, f, 3, d, True)
        conv(64, 64, 3, d, True)

    def synthesis():
        d = res // 8
        for f_in, f in ((64, 64), (64, 32), (32, 16)):
            conv(f_in, f, 3, d)                               # stride-2 transposed conv: 27 taps per INPUT voxel
            d *= 2
            conv(f, f, 3, d, True); conv(f, f, 3, d, True)
        conv(16, 1, 3, d)

    def hyper_a():
        d = res // 8
        conv(64, 64, 3, d, True); conv(64, 64, 3, d // 2); conv(64, 64, 3, d // 2, True)

    def hyper_s():
        d = res // 16
        conv(64, 64, 3, d, True); conv(64, 64, 3, d); conv(64, 64, 3, 2 * d, True)

    analysis(); hyper_a(); hyper_s(); synthesis()             # compress graph (model_types.py:379-388)
    hyper_s(); synthesis()                                    # decompress graph (:403-408)
    return alg, ex


def step_layers(res, batch):
    """Every conv launch of one c3p step (compress graph + decompress graph of one chunk), in the order of
    /root/reference/src/model_transforms.py:112-158: (transform id, layer index, calls per step, pcc_conv_desc fields, has residual)."""
    from pcc_geo_cnn_v2_amd import _lib as L
    rows = []

    def block(tid, first, d_in, specs, calls):
        d = d_in
        for i, (cin, cout, k, s, tr, res_add) in enumerate(specs):
            rows.append(dict(transform=tid, layer=first + i, calls=calls, cin=cin, cout=cout, k=k, stride=s, transposed=tr, d_in=d, residual=res_add))
            d = d * s if tr else -(-d // s)
        return d

    def blk(f_in, f, tr):
        return [(f_in, f, 3, 2, tr, False), (f, f, 3, 1, tr, False), (f, f, 3, 1, tr, True)]
    block(L.PCC_NET_ANALYSIS_PROGRESSIVE_V2, 0, res, blk(1, 16, 0) + blk(16, 32, 0) + blk(32, 64, 0) + [(64, 64, 3, 1, 0, False)], 1)
    block(L.PCC_NET_HYPER_ANALYSIS, 0, res // 8, [(64, 64, 3, 1, 0, False), (64, 64, 3, 2, 0, False), (64, 64, 3, 1, 0, False)], 1)
    block(L.PCC_NET_HYPER_SYNTHESIS, 0, res // 16, [(64, 64, 3, 1, 1, False), (64, 64, 3, 2, 1, False), (64, 64, 3, 1, 1, False)], 2)
    block(L.PCC_NET_SYNTHESIS_PROGRESSIVE_V2, 0, res // 8, blk(64, 64, 1) + blk(64, 32, 1) + blk(32, 16, 1) + [(16, 1, 3, 1, 1, False)], 2)
    return rows


def layer_roofs(ctx, row, batch, ms):
    """One row of roofline.step.layers: the layer's algorithmic HBM bytes (fp32 input + output + residual, unfused) and the multiply-adds
    its kernel family executes, each against the roof of the unit that does it, for a measured launch time."""
    import ctypes
    from pcc_geo_cnn_v2_amd import _lib as L
    d_in, cin, cout, k, s, tr = row['d_in'], row['cin'], row['cout'], row['k'], row['stride'], row['transposed']
    d_out = d_in * s if tr else -(-d_in // s)
    desc = L.ConvDesc(N=batch, D=d_in, H=d_in, W=d_in, Cin=cin, Cout=cout, k=k, stride=s, transposed=tr,
                      flags=L.PCC_CONV_BIAS | L.PCC_CONV_RELU | (L.PCC_CONV_ADD if row['residual'] else 0))
    buf = ctypes.create_string_buffer(96)
    L.check(L.lib().pcc_conv_kernel_family(ctx.handle, ctypes.byref(desc), buf, 96), 'pcc_conv_kernel_family')
    fam = buf.value.decode()
    grid = d_in if (tr and s == 2) else d_out                    # the grid the 27 taps are applied on
    macs = float(batch) * grid ** 3 * k ** 3 * cin * cout
    nbytes = 4.0 * batch * (d_in ** 3 * cin + d_out ** 3 * cout * (2 if row['residual'] else 1))
    ex = macs
    if fam.startswith('conv16_wino'):
        ex = macs * wino_exec_factor(16 if 'f16s' in fam else cin, d_in, batch, ctx.num_cu)
    if 'fp16 x 2' in fam:
        pipe, peak, mult = 'f16 MFMA (4 product terms per fp32 multiply-add)', PEAK_BF16_MFMA, 4.0
    elif 'bf16 x 3' in fam:
        pipe, peak, mult = 'bf16 MFMA (6 product terms per fp32 multiply-add)', PEAK_BF16_MFMA, 6.0
    elif 'VALU' in fam or 'generic' in fam:
        pipe, peak, mult = 'fp32 VALU', PEAK_FP32_MFMA, 1.0
    else:
        pipe, peak, mult = 'fp32 MFMA', PEAK_FP32_MFMA, 1.0
    t = ms * 1e-3
    hbm_frac = nbytes / t / 1e9 / PEAK_HBM
    pipe_frac = 2.0 * ex * mult / t / 1e12 / peak
    return dict(kernel_family=fam, algorithmic_bytes=nbytes, hbm_frac_of_8tbs=hbm_frac, pipe=pipe, executed_pipe_flops=2.0 * ex * mult,
                pipe_frac_of_peak=pipe_frac, frac_of_own_roof=max(hbm_frac, pipe_frac), bound='hbm' if hbm_frac >= pipe_frac else 'mfma'), nbytes, 2.0 * ex * mult, peak


def synthetic_weights(model, seed=42):
    from pcc_geo_cnn_v2_amd.entropy_models import EntropyBottleneck
    w = {k: v for k, v in model.get_weights().items() if not k.startswith('entropy_bottleneck/')}
    rng = np.random.default_rng(seed + 1)
    for k in list(w):
        if k.endswith('/kernel'):
            g = GAIN_ANALYSIS if k.startswith(('analysis/', 'hyper_')) else GAIN_SYNTHESIS
            w[k] = (w[k] * g).astype(np.float32)
        elif k.endswith('/bias'):
            w[k] = rng.normal(0, 0.05, w[k].shape).astype(np.float32)
    last = max(int(k.split('/')[1]) for k in w if k.startswith('synthesis/'))
    w[f'synthesis/{last}/bias'] = np.array([FINAL_BIAS], np.float32)
    # factorized prior: a fixed narrow table (an untrained tfc init_scale=10 prior would spend 5 bits on every z)
    for k, v in EntropyBottleneck.init_params(model.num_filters, init_scale=EB_INIT_SCALE, seed=seed).items():
        w[f'entropy_bottleneck/{k}'] = v
    return w


def synthetic_blocks(n, device, seed0):
    """Surface-like occupancy: thin shells of a seeded smooth random field (~1.5-3 % occupancy) -- SURVEY.md §8d."""
    out = torch.empty((n, RES, RES, RES), dtype=torch.float32, device=device)
    ax = torch.arange(RES, dtype=torch.float32, device=device)
    gx, gy, gz = torch.meshgrid(ax, ax, ax, indexing='ij')
    for i in range(n):
        g = torch.Generator(device='cpu').manual_seed(1234 + seed0 + i)
        field = torch.zeros((RES, RES, RES), device=device)
        for _ in range(4):
            c = (torch.rand(3, generator=g) * RES).to(device)
            r = float(torch.rand(1, generator=g)) * 20 + 8
            field += torch.exp(-((gx - c[0]) ** 2 + (gy - c[1]) ** 2 + (gz - c[2]) ** 2) / (2 * r * r))
        level = float(torch.quantile(field.flatten()[::64], 0.6))
        out[i] = ((field - level).abs() < 0.012).float()
    return out


def kernel_source_sha1(path):
    """git blob hash of a source file (what `git hash-object` prints): ties a profiled number to the code it was measured on."""
    import hashlib
    data = open(path, 'rb').read()
    return hashlib.sha1(b'blob %d\0' % len(data) + data).hexdigest()


def traffic_provenance():
    """HBM bytes per launch of the dominant kernel as profiled (rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE in separate passes,
    tools/profile_round.sh -> profiles/dominant_kernel_traffic.json), with the hash of the kernel source it was taken on and
    whether the source changed since (tests/test_round4_cpu.py fails on a stale file)."""
    prof = os.path.join(ROOT, 'profiles', 'dominant_kernel_traffic.json')
    if not os.path.exists(prof):
        return None
    d = json.load(open(prof))
    src = os.path.join(ROOT, 'pcc_geo_cnn_v2_amd', 'csrc', d.get('kernel_source', 'conv_wino.hip'))
    now = kernel_source_sha1(src) if os.path.exists(src) else None
    return {'file': 'profiles/dominant_kernel_traffic.json', 'kernel': d.get('kernel'), 'hbm_bytes_per_launch': d.get('hbm_bytes_per_launch'),
            'traffic_over_algorithmic': d.get('traffic_over_algorithmic'), 'kernel_source': d.get('kernel_source'),
            'kernel_source_sha1_at_measurement': d.get('kernel_source_sha1'), 'kernel_source_sha1_now': now,
            'stale': d.get('kernel_source_sha1') != now,
            'method': 'rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE, separate passes, on tools/bench_one.py (a separate profiled run, not this one)'}


def throttled_periods():
    """Scheduler periods in which this container was paused for exceeding its CPU quota (cgroup v2 cpu.stat), or None."""
    try:
        for line in open('/sys/fs/cgroup/cpu.stat'):
            if line.startswith('nr_throttled'):
                return int(line.split()[1])
    except OSError:
        pass
    return None


def cpu_baseline(model, w, blocks_np, budget_s=12.0):
    """The oracle's PyTorch-CPU port of the reference's batch-1 per-block loop, on the host cores."""
    from oracle import oracle as O
    from oracle import torch_oracle as T
    eb, gc = model.entropy_bottleneck, model.conditional_bottleneck
    om = dict(config='c3p', params=w, round_mode=0, data_format=model.data_format,
              eb=dict(cdf=eb.quantized_cdf, cdf_size=eb.cdf_length, offset=eb.offset, medians=eb.medians),
              gc=(gc.quantized_cdf, gc.cdf_length, gc.offset), scale_table=gc.scale_table_f32)
    T.codec_block_roundtrip(om, blocks_np[0][None, ..., None])  # warm-up (oneDNN primitive creation)
    # batch-1 convs do not scale to every core of a big host: calibrate the intra-op thread count on one block each
    # and time the sample with the fastest setting (reported as `cores`)
    cand = sorted({t for t in (8, 16, 32, 64, torch.get_num_threads()) if t <= torch.get_num_threads()})
    best_t, best = cand[-1], float('inf')
    for t in cand:
        torch.set_num_threads(t)
        t0 = time.perf_counter()
        T.codec_block_roundtrip(om, blocks_np[0][None, ..., None])
        dt = time.perf_counter() - t0
        if dt < best:
            best, best_t = dt, t
    torch.set_num_threads(best_t)
    n, t0 = 0, time.perf_counter()
    while True:
        T.codec_block_roundtrip(om, blocks_np[n % len(blocks_np)][None, ..., None])
        n += 1
        el = time.perf_counter() - t0
        if (el > budget_s and n >= 2) or n >= 1024:
            break
    return dict(value=n / el, unit='blocks/s', cores=torch.get_num_threads(), threads_used=torch.get_num_threads(), kind='port',
                host_logical_cores=os.cpu_count(), cpu_quota_cores=ops.usable_cores(),
                # the port follows the reference's encoder graph literally (model_types.py:383,387: it range-DECODES the strings it just
                # wrote to obtain z_hat / y_hat); the GPU path takes them from the quantiser (identical by construction), so the CPU
                # figure carries work the GPU figure legitimately does not
                includes_encoder_side_decode=True,
                note='cores = threads_used = oneDNN intra-op threads, calibrated on one block among the counts the container\'s CPU '
                     'quota allows (batch-1 convs stop scaling well below the core count); host_logical_cores = what the box shows',
                sample=f'{n} c3p 64^3 blocks, batch 1 (model_types.py:192-198 loop), oracle/torch_oracle.py: '
                       f'PyTorch-CPU oneDNN fp32 convs + C range coder, {el:.1f} s')


def standin_cloud():
    """The synthetic stand-in for longdress_vox10_1300 the tests use (tests/test_codec_gpu.py): a thin shell at 1024^3, ~5e5 points,
    ~190 occupied 64^3 blocks at octree level 4.  Seeded: the same cloud on every rank."""
    rng = np.random.default_rng(0)
    u = rng.standard_normal((3_000_000, 3))
    return np.unique(np.round(u / np.linalg.norm(u, axis=1, keepdims=True) * 200 + np.array([512, 500, 520])).astype(np.int64), axis=0).astype(np.float64)


def run_configs2(args, dist, rank, world, device, ctx, coder_threads):
    """BASELINE.json configs[2]: ONE octree-split cloud through compress_blocks / decompress_blocks (the calls of
    /root/reference/src/compress_octree.py:97-105 and decompress_octree.py:60-66), its blocks sharded over the ranks (sharding.py), the
    closing collectives INSIDE the clock.  Strong scaling: the cloud is fixed, `value` = blocks of the cloud x steps / max-over-ranks time.
    A step = encode (partitioned blocks on the host -> container bytes on rank 0) + the hand-over of the container (rank 0 writes a file,
    every rank reads it, as the two CLIs do) + decode (-> points on rank 0)."""
    import gzip
    import io
    import tempfile
    from pcc_geo_cnn_v2_amd import model_syntax, sharding
    from pcc_geo_cnn_v2_amd.utils.octree_coding import partition_octree
    R, level, res = 1024, 4, 64
    longdress = os.environ.get('PCC_BENCH_CLOUD')           # a real vox10 .ply when present (no dataset ships with the container)
    if longdress:
        from pcc_geo_cnn_v2_amd.utils import pc_io
        pts = pc_io.load_points([longdress])[0].astype(np.float64)
    else:
        pts = standin_cloud()
    blocks, binstr = partition_octree(pts, [0, 0, 0], [R] * 3, level)
    enc = ModelConfigType['c3p'].build(batch_size=args.chunk, coder_threads=coder_threads)
    enc.compress([1, 1, res, res, res])
    w = synthetic_weights(enc)
    enc.set_weights(w)
    dec = ModelConfigType['c3p'].build(batch_size=args.chunk, coder_threads=coder_threads)
    dec.decompress()
    dec.set_weights({k: v for k, v in enc.get_weights().items() if not k.startswith(('analysis/', 'hyper_analysis/'))})
    path = os.path.join(tempfile.gettempdir(), f'pcc_bench_cfg2_{os.environ.get("MASTER_PORT", "0")}.bin')
    phase = {'encode': 0.0, 'handover': 0.0, 'decode': 0.0}

    def barrier():
        torch.cuda.synchronize(device)
        if dist is not None:
            dist.barrier()

    def step():
        t0 = time.perf_counter()
        streams, infos, _ = enc.compress_blocks(ctx, blocks, binstr, pts, R, level, opt_metrics=('d1_mse',), fixed_threshold=args.fixed_threshold, need_points=False)
        t1 = time.perf_counter()
        if rank == 0:
            raw = model_syntax.save_compressed_file(binstr, streams[0], R, level, strict=True)
            with open(path + '.tmp', 'wb') as fh:
                fh.write(gzip.compress(raw, 6))
            os.replace(path + '.tmp', path)
        if dist is not None:
            dist.barrier()          # the decoder processes start after the encoder wrote the file
        with gzip.open(path, 'rb') as fh:
            _, _, binstr2, data2 = model_syntax.load_compressed_file(fh)
        t2 = time.perf_counter()
        out, _ = dec.decompress_blocks(ctx, data2, [res] * 3)
        t3 = time.perf_counter()
        phase['encode'] += t1 - t0; phase['handover'] += t2 - t1; phase['decode'] += t3 - t2
        return (sum(len(b) for b in out) if out is not None else 0), os.path.getsize(path)

    for _ in range(max(args.warmup, 2)):
        step()
    for k in phase:
        phase[k] = 0.0
    sharding.stats_begin()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        n_pts, n_bytes = step()
    barrier()
    elapsed = time.perf_counter() - t0
    coll = sharding.stats_end()
    if dist is not None:
        tt = torch.tensor([elapsed, coll['seconds']], dtype=torch.float64, device=device)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed, coll_s = float(tt[0].item()), float(tt[1].item())
    else:
        coll_s = coll['seconds']
    if rank == 0:
        try:
            os.remove(path)
        except OSError:
            pass
        out = {'metric': 'voxel_blocks_64cubed_per_sec_encode_decode_ONE_CLOUD_SHARDED_configs2_not_the_headline', 'value': len(blocks) * args.steps / elapsed, 'unit': 'blocks/s',
               'n_gpus': world, 'steps': args.steps, 'warmup': max(args.warmup, 2), 'ms_per_step': 1e3 * elapsed / args.steps, 'higher_is_better': True,
               'scaling': 'strong', 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
               'config': {'workload': ('BASELINE.json configs[2]: c3p, one vox10-sized cloud octree-split (level 4) into 64^3 blocks, sharded over the ranks in contiguous '
                                       'Morton ranges, ' + ('fixed threshold' if args.fixed_threshold else 'adaptive threshold search on d1_mse (the CLI default)') + ', compress_blocks + container hand-over + decompress_blocks; '
                                       + ('cloud = ' + os.path.basename(longdress) if longdress else 'cloud = seeded thin-shell stand-in for longdress_vox10_1300 (no dataset in the container)')),
                          'points': int(len(pts)), 'blocks': len(blocks), 'blocks_per_rank': sharding.shard_sizes(len(blocks), world), 'pipeline_chunk': args.chunk,
                          'container_bytes_gzip': int(n_bytes), 'decoded_points': int(n_pts), 'codec_numerics': ctx.numerics_tag('fp32'),
                          'phase_ms_per_step_rank0': {k: 1e3 * v / args.steps for k, v in phase.items()},
                          'what_a_step_includes': 'host: dense-block upload, range coder, whole-cloud D1 metrics of the selected reconstruction (select_best_per_opt_metric: KD-trees '
                                                  'on the host, the original cloud is replicated), container serialisation + gzip, file write / read; GPU: both graphs and the threshold search; '
                                                  'collectives: sharding.py (2 per cloud on the encoder, 2 on the decoder)'},
               'final_collectives_ms': 1e3 * coll_s / args.steps, 'final_collectives_calls_per_step': coll['calls'] / args.steps,
               'final_collectives_bytes_sent_per_step_rank0': coll['bytes'] / args.steps,
               'final_collectives_note': 'max over ranks of the host wall time inside the collectives of sharding.py (call -> result on the host), inside the clock; 0 at world 1',
               'roofline': None, 'cpu_baseline': None}
        if dist is not None:
            out['multi_gpu'] = {'rccl_world_size': dist.get_world_size(), 'backend': dist.get_backend()}
        import ctypes
        ctypes.CDLL(None).fflush(None)
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    return 0


def self_launch(n):
    import socket
    import subprocess
    assert torch.cuda.device_count() >= n or '--dry-run' in sys.argv, \
        f'--gpus {n} but only {torch.cuda.device_count()} GPU(s) visible: refusing to run a mislabeled bench'
    sock = socket.socket()
    sock.bind(('127.0.0.1', 0))
    port = sock.getsockname()[1]
    sock.close()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'))
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={n}', '--master-addr', '127.0.0.1',
           '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def dry_run(args, dist, rank, world):
    """The measurement skeleton without the GPU work: barrier, timed region, MAX over ranks, SUM of the per-rank block counts,
    one JSON line from rank 0."""
    def barrier():
        if dist is not None:
            dist.barrier()
    barrier()
    t0 = time.perf_counter()
    time.sleep(0.002 * args.steps)
    n_blocks = args.steps * BATCH
    barrier()
    elapsed = time.perf_counter() - t0
    tot = n_blocks
    if dist is not None:
        tt = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
        st = torch.tensor([n_blocks], dtype=torch.float64)
        dist.all_reduce(st)
        tot = int(st.item())
    if rank == 0:
        print(json.dumps({'metric': 'DRY_RUN_launcher_skeleton_only', 'dry_run': True, 'value': tot / elapsed, 'unit': 'blocks/s',
                          'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'blocks_total': tot}), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=200, help='timed steps (default 200: about one second, so that pipeline fill and drain -- two chunks -- weigh 1 %%)')
    ap.add_argument('--warmup', type=int, default=10)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-ab', action='store_true', help='skip the exact-fp32 A/B (>= 40 steps with the split-bf16 kernels off) that the default run appends')
    ap.add_argument('--no-secondary', action='store_true', help='skip the configs[4] (128^3, batch 8, fp16) measurement that the default run appends')
    ap.add_argument('--chunk', type=int, default=CHUNK)
    ap.add_argument('--coder-threads', type=int, default=0,
                    help='host range-coder threads of this rank (default: logical cores / ranks); for studying the host share')
    ap.add_argument('--precision', default='fp32', choices=['fp32', 'fp16'],
                    help="fp32 = the reference's arithmetic (the headline number); fp16 = fp16 MFMA / fp32 accumulate (BASELINE.json configs[4] flavour, informational)")
    ap.add_argument('--workload', default='configs1', choices=['configs1', 'configs4', 'configs2'],
                    help='configs1 = BASELINE.json configs[1] (c3p, batch 32, 64^3: the headline); configs4 = configs[4] (deepest config, '
                         '128^3 blocks, batch 8, fp16 MFMA): a SEPARATE, labelled line, never the headline; configs2 = configs[2] (one octree-split cloud '
                         'sharded over the ranks, strong scaling, closing collectives inside the clock): also a separate, labelled line')
    ap.add_argument('--fixed-threshold', action='store_true', help='configs2: --fixed_threshold of the CLI instead of the adaptive search')
    ap.add_argument('--dry-run', action='store_true',
                    help='launcher / collective skeleton only (gloo on CPU, no GPU work, value is meaningless): used by the CPU tests')
    args = ap.parse_args()

    global RES, BATCH, FLOPS_PER_BLOCK
    if args.workload == 'configs4':
        RES, BATCH, FLOPS_PER_BLOCK = 128, 8, 248.689e9          # c3p @128^3
        args.precision, args.chunk = 'fp16', min(args.chunk, 8)
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        # bare `python bench.py --gpus N`: re-exec under torch.distributed.run (one rank per GPU); never a mislabeled 1-GPU run
        return self_launch(args.gpus)
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    dist = None
    if world > 1 or os.environ.get('PCC_BENCH_FORCE_DIST'):     # (the knob exercises the RCCL path on a 1-GPU box)
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if world == 1:      # the knob without a launcher: a one-rank group
            for k, v in (('RANK', '0'), ('WORLD_SIZE', '1'), ('LOCAL_RANK', '0'), ('MASTER_PORT', '29517')):
                os.environ.setdefault(k, v)
        if args.dry_run:
            dist.init_process_group('gloo')
        else:
            torch.cuda.set_device(local_rank)
            dist.init_process_group('nccl', device_id=torch.device('cuda', local_rank))
    assert world == max(args.gpus, 1), f'--gpus {args.gpus} but WORLD_SIZE {world}: one rank per GPU'
    if args.dry_run:
        return dry_run(args, dist, rank, world)
    assert torch.cuda.device_count() > local_rank, \
        f'rank {rank} wants GPU {local_rank} but only {torch.cuda.device_count()} are visible: refusing to run a mislabeled bench'
    device = torch.device('cuda', local_rank)
    torch.cuda.set_device(device)
    ctx = ops.get_context(device)
    # host threads follow the cores the container may really use (its CPU quota), not the cores it can see
    torch.set_num_threads(max(1, min(torch.get_num_threads(), ops.usable_cores() // max(world, 1))))

    # host range-coder threads: share the node's cores between the ranks
    # (cgroup-quota aware: a 16-CPU container that shows 256 cores gets throttled by 256 threads; never more threads than the
    # cores this rank owns: 8 ranks in a 16-core container get 2 each)
    cores_per_rank = max(1, ops.usable_cores() // max(world, 1))
    coder_threads = args.coder_threads or cores_per_rank
    if args.workload == 'configs2':
        return run_configs2(args, dist, rank, world, device, ctx, coder_threads)
    model = ModelConfigType['c3p'].build(batch_size=args.chunk, coder_threads=coder_threads, precision=args.precision)
    model.compress([1, 1, RES, RES, RES])
    w = synthetic_weights(model)
    model.set_weights(w)

    # every rank codes its own shard of independent blocks (weak scaling: BATCH blocks per GPU per step)
    x = synthetic_blocks(BATCH, device, seed0=rank * BATCH)
    chunks = [x[i:i + args.chunk].contiguous() for i in range(0, BATCH, args.chunk)]

    stamps = []          # host time at which the results of every chunk became available (steady-state window below)

    def run(steps, mdl=None, chs=None):
        n_pts, n_bytes, n_blocks = 0, 0, 0
        del stamps[:]
        for strings, cnt_e, pts in (mdl or model).roundtrip_stream(ctx, (c for _ in range(steps) for c in (chs or chunks))):
            stamps.append(time.perf_counter())
            n_blocks += len(strings)
            n_bytes += sum(len(s) for ss in strings for s in ss)
            n_pts += sum(len(p) for p in pts)
        return n_blocks, n_bytes, n_pts

    def steady_ms_per_step(chunks_per_step):
        """ms per step inside a window that starts after the 3-deep pipeline has filled and ends before it drains: the
        results of chunk k arrive while chunks k+1, k+2 are in flight, so the spacing of arrivals is the step time."""
        K = len(stamps)
        if K < 8:
            return None
        return 1e3 * (stamps[K - 3] - stamps[2]) / (K - 5) * chunks_per_step

    def barrier():
        torch.cuda.synchronize(device)
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(device)

    # Warm-up floor.  The first process on a fresh box pays one-time costs in its first ~10 steps -- first touch of the pinned ring
    # buffers (the pipeline is three chunks deep plus look-ahead), lazy code-object loads, CPU / GPU clock ramp (measured: arrival gaps
    # of 5.6 and 8.1 ms among 4.4 ms ones in a --steps 20 --warmup 3 run that was the first process of its box, none in the processes
    # after it).  The untimed steps are therefore max(W, 10) (round 4 ran 10 + W): with the default W = 10 nothing is added, with a
    # smaller W the difference runs first as set-up and is reported as config.setup_priming_steps.
    PRIME_STEPS = 0 if os.environ.get('PCC_BENCH_NO_PRIME') else max(0, 10 - max(args.warmup, 0))
    if PRIME_STEPS:
        run(PRIME_STEPS)
    run(max(args.warmup, 0)) if args.warmup > 0 else None

    # live HIP-event timing of the dominant kernel on its launch stream: Conv3DTranspose 16->16 k3 s1 @64^3 + residual = layer 8
    # of the c3p synthesis transform (its twin, layer 7, runs the same kernel without the residual read; together 47 % of all
    # MACs).  The library records the events around that layer inside pcc_codec_encode / pcc_codec_decode_main.
    from pcc_geo_cnn_v2_amd import _lib as L
    DOM_LAYER = 8
    # every PROFILE_STRIDE-th launch of the layer is timed: an event record costs the queue ~6 us, four of them per step would be
    # 0.5 % of what is being measured
    ops.profile_select(ctx, L.PCC_NET_SYNTHESIS_PROGRESSIVE_V2, DOM_LAYER, stride=PROFILE_STRIDE)
    # The interpreter holds ~1 M long-lived objects by now (torch, numpy, the package); a generation-2 collection that walks them takes
    # 5 - 17 ms -- one landing inside a 77 ms timed region is the sporadic chunk gap of the driver's 20-step runs (1 run in 5).  Collect
    # now and move the survivors to the permanent generation (gc.freeze: what long-running Python services do after start-up): the
    # collector stays ON, later collections only walk what the steps allocate.  PCC_BENCH_GC_DEFAULT=1: leave the collector as it is.
    import gc
    if os.environ.get('PCC_BENCH_NOGC'):
        gc.collect(); gc.disable()
    elif not os.environ.get('PCC_BENCH_GC_DEFAULT'):
        gc.collect(); gc.freeze()
    barrier()
    dev_allocs0 = torch.cuda.memory_stats(device).get('num_device_alloc', 0)
    cpu0 = time.process_time()
    thr0 = throttled_periods()
    t0 = time.perf_counter()
    n_blocks, n_bytes, n_pts = run(args.steps)
    barrier()
    thr1 = throttled_periods()
    host_cores_busy = (time.process_time() - cpu0) / (time.perf_counter() - t0)       # CPU time of all threads of this rank / wall time
    dev_allocs = torch.cuda.memory_stats(device).get('num_device_alloc', 0) - dev_allocs0      # hipMalloc calls inside the timed region (each one stalls the queue)
    if os.environ.get('PCC_BENCH_STAMPS'):      # arrival spacing of the chunks (ms), for pipeline debugging
        print('device allocations inside the timed region:', dev_allocs, file=sys.stderr)
        print('stamps', ' '.join(f'{1e3 * (b - a):.2f}' for a, b in zip(stamps[:-1], stamps[1:])), file=sys.stderr)
    t1 = time.perf_counter()
    elapsed = t1 - t0
    steady_ms = steady_ms_per_step(BATCH // args.chunk)
    kern_ms = ops.profile_read(ctx)
    ops.profile_select(ctx, -1, -1)
    n_launch = 2 * args.steps * (BATCH // args.chunk)
    assert len(kern_ms) == (n_launch + PROFILE_STRIDE - 1) // PROFILE_STRIDE, f'{len(kern_ms)} timed launches of the dominant layer'
    multi = None
    if dist is not None:
        # what the driver cannot see from outside: that RCCL really spans `world` ranks on distinct devices, what each rank
        # did, and what the closing collectives cost
        own = torch.tensor([elapsed, n_blocks, float(steady_ms or 0.0), host_cores_busy], dtype=torch.float64, device=device)
        torch.cuda.synchronize(device)
        tc0 = time.perf_counter()
        tt = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
        stats = torch.tensor([n_blocks, n_bytes, n_pts], dtype=torch.float64, device=device)
        dist.all_reduce(stats)  # the one collective of the sharded path: totals to rank 0
        tot_blocks = int(stats[0].item())
        torch.cuda.synchronize(device)
        coll_ms = 1e3 * (time.perf_counter() - tc0)
        per_rank = [torch.zeros_like(own) for _ in range(dist.get_world_size())]
        dist.all_gather(per_rank, own)
        uuid = str(getattr(torch.cuda.get_device_properties(device), 'uuid', f'cuda:{local_rank}'))
        uuids = [None] * dist.get_world_size()
        dist.all_gather_object(uuids, uuid)
        multi = {'rccl_world_size': dist.get_world_size(), 'backend': dist.get_backend(), 'device_uuids': uuids,
                 'distinct_devices': len(set(uuids)),
                 'per_rank_blocks_per_s': [float(r[1] / r[0]) for r in per_rank],
                 'per_rank_elapsed_s': [float(r[0]) for r in per_rank],
                 'per_rank_steady_ms_per_step': [float(r[2]) or None for r in per_rank],
                 'host_cores_busy_per_rank': [round(float(r[3]), 2) for r in per_rank], 'host_cores_per_rank': cores_per_rank,
                 'host_bound': bool(max(float(r[3]) for r in per_rank) >= 0.9 * cores_per_rank),
                 'final_collectives_ms': coll_ms,
                 'final_collectives': 'all_reduce(MAX) of the elapsed time + all_reduce(SUM) of (blocks, bytes, points); after the timed region'}
    else:
        tot_blocks = n_blocks
    assert n_blocks == args.steps * BATCH

    # Same-process A/B under the same clock: the exact-fp32 MFMA kernels everywhere (the round-3 numerics; PCC_NO_SPLIT=1 selects them
    # at start-up, here the context's numerics word is switched -- encoder and decoder of a chunk share the context, so they agree).
    ab_exact = None
    if args.workload == 'configs1' and args.precision == 'fp32' and world == 1 and not args.no_ab and not (ctx.numerics()[1] & L.PCC_NUM['no_split']):
        ab_steps = max(40, min(args.steps, 100))
        with ctx.numerics_override(no_split=True):
            run(3)
            ops.profile_select(ctx, L.PCC_NET_SYNTHESIS_PROGRESSIVE_V2, DOM_LAYER, stride=PROFILE_STRIDE)
            torch.cuda.synchronize(device)
            a0 = time.perf_counter()
            nb_ab, _, _ = run(ab_steps)
            torch.cuda.synchronize(device)
            a1 = time.perf_counter()
            k_ab = ops.profile_read(ctx)
            ops.profile_select(ctx, -1, -1)
        ab_exact = {'what': 'the same workload in the same process with every split-bf16 kernel off (numerics switch no_split = PCC_NO_SPLIT=1: exact-fp32 '
                            'v_mfma_f32_16x16x4_f32 everywhere, the round-3 kernels)', 'value': nb_ab / (a1 - a0), 'unit': 'blocks/s', 'steps': ab_steps, 'warmup': 3,
                    'ms_per_step': 1e3 * (a1 - a0) / ab_steps, 'steady_ms_per_step': steady_ms_per_step(BATCH // args.chunk),
                    'dominant_kernel': 'conv16_wino_kernel<relu> (exact fp32 MFMA Winograd), synthesis layer 8',
                    'dominant_avg_launch_ms': float(np.mean(k_ab)) if k_ab else None,
                    'dominant_median_launch_ms': float(np.median(k_ab)) if k_ab else None, 'launches_timed': len(k_ab)}

    # BASELINE.json configs[4] (c6 = the c3p graph, 128^3 blocks, batch 8, fp16 MFMA) measured in the SAME process, so that the
    # driver's one `bench.py --gpus 1` run times it too.  A separate, labelled object: never the headline value.
    secondary = None
    if args.workload == 'configs1' and args.precision == 'fp32' and world == 1 and not args.no_secondary:
        res2, batch2, steps2 = 128, 8, max(40, args.steps // 2)      # >= 40 steps: pipeline fill and drain (two chunks) weigh 5 % at most
        m2 = ModelConfigType['c3p'].build(batch_size=batch2, coder_threads=coder_threads, precision='fp16')
        m2.compress([1, 1, res2, res2, res2])
        m2.set_weights(w)
        g = torch.Generator(device='cpu').manual_seed(99)
        x2 = (torch.rand((batch2, res2, res2, res2), generator=g) < 0.02).float().to(device)     # Bernoulli(0.02) occupancy (SURVEY.md 8d variant)
        ch2 = [x2]
        run(2, m2, ch2)
        ops.profile_select(ctx, L.PCC_NET_SYNTHESIS_PROGRESSIVE_V2, DOM_LAYER, stride=PROFILE_STRIDE)
        torch.cuda.synchronize(device)
        th2 = throttled_periods()
        s0 = time.perf_counter()
        nb2, nby2, npt2 = run(steps2, m2, ch2)
        torch.cuda.synchronize(device)
        el2 = time.perf_counter() - s0
        th2 = None if th2 is None else throttled_periods() - th2
        k2 = ops.profile_read(ctx)
        ops.profile_select(ctx, -1, -1)
        avg2 = float(np.mean(k2)) if k2 else float('nan')
        bytes2 = 3.0 * batch2 * res2 ** 3 * 16 * 2
        secondary = {'metric': 'voxel_blocks_128cubed_per_sec_encode_decode_FP16_MODE', 'value': nb2 / el2, 'unit': '128^3 blocks/s',
                     'equivalent_64cubed_blocks_per_s': 8 * nb2 / el2, 'steps': steps2, 'warmup': 2, 'ms_per_step': 1e3 * el2 / steps2,
                     'steady_ms_per_step': steady_ms_per_step(1), 'cpu_quota_throttled_periods': th2,
                     'dtype': 'f16 operands and mid-network storage / f32 accumulate', 'data': 'synthetic',
                     'config': {'workload': 'BASELINE.json configs[4]: deepest config (paper c6 = the c3p graph), batch=8 synthetic 128^3 occupancy grids, '
                                            'fp16 MFMA with fp16 mid-network storage, fixed threshold idx 128, encode+decode',
                                'bytes_per_block': nby2 / nb2, 'decoded_points_per_block': npt2 / nb2},
                     'roofline': {'bound': 'hbm', 'kernel': 'conv_f16_kernel<16>: Conv3DTranspose 16->16 k3 s1 + fp16 residual @128^3 x8 (synthesis layer 8)',
                                  'achieved': bytes2 / (avg2 * 1e-3) / 1e9, 'peak': 8000.0, 'unit': 'GB/s', 'frac': bytes2 / (avg2 * 1e-3) / 1e9 / 8000.0,
                                  'traffic': None, 'algorithmic_bytes_per_launch': bytes2, 'avg_launch_ms': avg2, 'launches_timed': len(k2)}}
        del m2, x2

    # roofline.step: where the whole step is.  Every conv layer of the four transforms is timed in turn with the
    # library's HIP-event hook (one layer per pass of 3 steps, after the timed region, same process, same clock state as the pipeline
    # keeps running) and priced against its own roof; what the list does not cover (quantisers, packing, threshold + compaction,
    # the host coder's shadow, queue gaps, minus the overlap of the encoder and decoder streams) is the residual row.
    step_profile = None
    if world == 1 and args.precision == 'fp32' and args.workload == 'configs1' and not os.environ.get('PCC_BENCH_NO_STEP_PROFILE'):
        rows = step_layers(RES, args.chunk)
        for r in rows:
            ops.profile_select(ctx, r['transform'], r['layer'], stride=1)
            run(3)
            torch.cuda.synchronize(device)
            k_ms = ops.profile_read(ctx)
            ops.profile_select(ctx, -1, -1)
            r['launch_ms'] = [float(v) for v in k_ms]
        step_profile = rows

    if rank == 0:
        value = tot_blocks / elapsed
        flops_launch = 2.0 * args.chunk * RES ** 3 * 27 * 16 * 16     # algorithmic flops of one dominant launch
        avg_ms = float(np.mean(kern_ms)) if kern_ms else float('nan')
        achieved = flops_launch / (avg_ms * 1e-3) / 1e12
        num_sw = ctx.numerics()[1]
        winograd = not (num_sw & L.PCC_NUM['no_winograd'])
        split = winograd and not (num_sw & L.PCC_NUM['no_split'])      # 16-channel layers on the bf16 MFMA pipe (conv_wino_bf16.hip)
        alg_bytes_launch = 3.0 * args.chunk * RES ** 3 * 16 * 4           # input + residual + output of the timed layer, fp32
        if winograd:
            # F(2x2,3x3) in x-y (16 instead of 36 multiplies per 2x2 outputs and z tap); padding planes skipped
            exec_flops = flops_launch * wino_exec_factor(16, RES, args.chunk, ctx.num_cu)
            f16s = split and not (num_sw & L.PCC_NUM['no_f16s'])
            if f16s:
                dom_kernel = ('conv16_wino_f16s_kernel<relu, G = 1> (Conv3DTranspose 16->16 k3 s1 @64^3 + residual: synthesis layer 8, timed in the encoder and in the '
                              'decoder; layer 7 runs the same kernel: 4 launches per step)')
                dom_note = ('two-piece fp16 Winograd (round 6): every fp32 operand = h + l fp16 pieces under an exact power-of-two pre-scale (weights: per layer; '
                            'activations: per 64^3 block, from the max |x| its producer recorded), all four product terms in two v_mfma_f32_16x16x32_f16 per row '
                            '(fp32 accumulate), operand error 2^-22 (tests/test_conv_gpu.py: the same gates as the exact-fp32 kernel).  Nearest hardware roof: HBM '
                            '(achieved/frac = algorithmic bytes: input + residual + output, / HIP-event launch time / 8 TB/s); `mfma` restates it against the matrix '
                            'peaks; PCC_NO_F16S=1 gives the three-piece bf16 kernel of round 4, PCC_NO_SPLIT=1 the exact-fp32 line')
            elif split:
                dom_kernel = ('conv16_wino_bf16_kernel<relu> (Conv3DTranspose 16->16 k3 s1 @64^3 + residual: synthesis layer 8, timed in the encoder and in the '
                              'decoder; layer 7 runs the same kernel: 4 launches per step)')
                dom_note = ('split-bf16 Winograd: every fp32 operand = three bf16 pieces, six product terms in three v_mfma_f32_16x16x32_bf16 per row (fp32 '
                            'accumulate), error equal to the exact-fp32 MFMA kernel (tests/test_conv_gpu.py).  The matrix work is 2.7x shorter than on the fp32 '
                            'pipe, so the launch is no longer MFMA-bound: the nearest hardware roof is HBM (achieved/frac = algorithmic bytes: input + residual '
                            '+ output, / HIP-event launch time / 8 TB/s); what actually limits it is VALU issue of the operand split at one wave per SIMD '
                            '(DESIGN_HISTORY.md 3.0c, profiles/r04_*).  `mfma` restates it against the matrix peaks; PCC_NO_SPLIT=1 gives the fp32 line')
            else:
                dom_kernel = 'conv16_wino_kernel<relu> (Conv3DTranspose 16->16 k3 s1 @64^3 + residual: synthesis layer 8, timed in the encoder and in the decoder; layer 7 runs the same kernel: 4 launches per step)'
                dom_note = ('achieved/frac = fp32 MFMA flops the kernel EXECUTES (Winograd F(2x2,3x3) in x-y + direct z: 16/36 * 63.67/64 of '
                            'the direct-convolution flops for a whole-volume slab) / HIP-event launch time / dense fp32 MFMA peak; algorithmic_* restate it in the '
                            'direct-convolution flops of SURVEY.md 8d (what a direct kernel would have to sustain for the same time)')
        else:
            exec_flops = flops_launch
            dom_kernel = 'conv16_pers_kernel<2,4,2,20,2> (Conv3DTranspose 16->16 k3 s1 @64^3, 4 launches per step)'
            dom_note = 'direct implicit-GEMM kernel (PCC_NO_WINOGRAD=1): executed == algorithmic flops'
        achieved_exec = exec_flops / (avg_ms * 1e-3) / 1e12
        alg_step, exec_step = c3p_step_flops(RES, args.chunk, num_cu=ctx.num_cu, winograd=winograd, split=split)
        assert abs(alg_step / FLOPS_PER_BLOCK - 1) < 2e-3, (alg_step, FLOPS_PER_BLOCK)       # the layer walk reproduces SURVEY.md 8d
        alg_bytes16 = 3.0 * args.chunk * RES ** 3 * 16 * 2            # fp16 mode: in + residual + out of the timed layer, fp16
        # HBM traffic of the dominant kernel is a PMC measurement of a separate profiled run (profiles/, tools/profile_round.sh): it is
        # reported as what it is -- `traffic` (this run) stays null
        traffic_profiled = traffic_provenance()
        if split:
            dom_roofline = {'bound': 'hbm', 'kernel': dom_kernel, 'achieved': alg_bytes_launch / (avg_ms * 1e-3) / 1e9, 'peak': PEAK_HBM, 'unit': 'GB/s',
                            'frac': alg_bytes_launch / (avg_ms * 1e-3) / 1e9 / PEAK_HBM, 'traffic': None, 'traffic_profiled': traffic_profiled,
                            'algorithmic_bytes_per_launch': alg_bytes_launch, 'avg_launch_ms': avg_ms, 'median_launch_ms': float(np.median(kern_ms)) if kern_ms else None,
                            'launches_timed': len(kern_ms), 'timed_launch_stride': PROFILE_STRIDE,
                            'frac_definition': ('HBM roof: algorithmic bytes of one launch (fp32 input + residual + output = 3 x 32 x 64^3 x 16 x 4 B) / mean HIP-event '
                                                'launch time / 8 TB/s.  NOT comparable with rounds 1-3, whose frac was executed fp32-MFMA flops / 157.3 TFLOP/s '
                                                '(that figure is kept as mfma.fp32_equivalent_over_fp32_mfma_peak; the bf16-pipe utilisation is '
                                                'mfma.frac_of_bf16_mfma_peak)'),
                            'mfma': {'executed_bf16_flops_per_launch': (4.0 if f16s else 6.0) * exec_flops, 'executed_bf16_tflops': (4.0 if f16s else 6.0) * achieved_exec,
                                     'frac_of_bf16_mfma_peak': (4.0 if f16s else 6.0) * achieved_exec / PEAK_BF16_MFMA,
                                     'fp32_equivalent_flops_per_launch': exec_flops, 'fp32_equivalent_tflops': achieved_exec,
                                     'fp32_equivalent_over_fp32_mfma_peak': achieved_exec / PEAK_FP32_MFMA,
                                     'algorithmic_flops_per_launch': flops_launch, 'algorithmic_tflops': achieved,
                                     'algorithmic_over_fp32_mfma_peak': achieved / PEAK_FP32_MFMA,
                                     'note': ('two f16 MFMAs of K = 32 replace four fp32 MFMAs of K = 4 per row: 4x the multiply-adds of the fp32 kernel (the executed_bf16_* keys hold f16-pipe figures: same 2.5 PF peak)'
                                              if f16s else 'three bf16 MFMAs of K = 32 replace four fp32 MFMAs of K = 4 per row: 6x the multiply-adds of the fp32 kernel, on a 16x faster pipe')},
                            'note': dom_note}
        else:
            dom_roofline = {'bound': 'mfma', 'kernel': dom_kernel,
                            'achieved': achieved_exec, 'peak': PEAK_FP32_MFMA, 'unit': 'TFLOP/s', 'frac': achieved_exec / PEAK_FP32_MFMA,
                            'traffic': None, 'traffic_profiled': traffic_profiled if winograd else None, 'algorithmic_bytes_per_launch': alg_bytes_launch,
                            'executed_flops_per_launch': exec_flops, 'avg_launch_ms': avg_ms, 'launches_timed': len(kern_ms), 'timed_launch_stride': PROFILE_STRIDE,
                            'frac_definition': 'fp32 MFMA roof: executed v_mfma_f32_16x16x4_f32 flops of one launch / mean HIP-event launch time / 157.3 TFLOP/s',
                            'algorithmic_flops_per_launch': flops_launch, 'algorithmic_tflops': achieved,
                            'algorithmic_speedup_vs_direct_roof': achieved / PEAK_FP32_MFMA,
                            'note': dom_note}
        out = {
            'metric': 'voxel_blocks_64cubed_per_sec_encode_decode' if args.workload == 'configs1' else
                      'voxel_blocks_128cubed_per_sec_encode_decode_FP16_MODE_not_the_headline', 'value': value, 'unit': 'blocks/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': 1e3 * elapsed / args.steps,
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32' if args.precision == 'fp32' else 'f16 operands / f32 accumulate (NOT the headline precision)', 'data': 'synthetic',
            'config': {'workload': ('c3p, lambda-independent graph, batch=32 synthetic 64^3 occupancy grids per GPU, '
                                    'fixed threshold idx 128, encode+decode (BASELINE.json configs[1])') if args.workload == 'configs1' else
                                   ('deepest config (paper c6 = the c3p graph), batch=8 synthetic 128^3 occupancy grids per GPU, fp16 MFMA with fp16 '
                                    'mid-network storage, fixed threshold idx 128, encode+decode (BASELINE.json configs[4])'),
                       'blocks_per_gpu_per_step': BATCH, 'pipeline_chunk': args.chunk, 'coder_threads_per_rank': coder_threads,
                       'host_cores_busy_per_rank': round(host_cores_busy, 2), 'host_cpu_quota_cores': ops.usable_cores(),
                       'host_cores_per_rank': cores_per_rank, 'host_bound': bool(host_cores_busy >= 0.9 * cores_per_rank),
                       'host_min_cores_per_rank_measured': HOST_MIN_CORES_PER_RANK,
                       'setup_priming_steps': PRIME_STEPS, 'python_gc': 'disabled' if os.environ.get('PCC_BENCH_NOGC') else 'default' if os.environ.get('PCC_BENCH_GC_DEFAULT') else 'on; collected and frozen (gc.freeze) after the warm-up', 'codec_numerics': ctx.numerics_tag(args.precision),
                       'device_allocations_in_timed_region': dev_allocs,
                       'cpu_quota_throttled_periods_in_timed_region': None if thr0 is None else thr1 - thr0, 'sharding': f'blocks x{world}',
                       'weights': f'synthetic Glorot-uniform, gains {GAIN_ANALYSIS}/{GAIN_SYNTHESIS}, seed 42',
                       'mfma_operand_split': (('Two forms, fp32 accumulate in a fixed order in both.  (a) two fp16 pieces (11 + 11 significand bits) under an exact '
                                               'power-of-two pre-scale -- weights per layer, activations per 64^3 block from the max |x| recorded by the producing kernel -- '
                                               'all four product terms in 2 x v_mfma_f32_16x16x32_f16 per product row: the Winograd layers, 16 -> 16 @64^3 / @32^3 (6 launches '
                                               'per 32-block step) and 32 -> 32 @32^3 (4) (conv_wino_f16s.hip, round 6).  (b) three bf16 pieces (8 + 8 + 8 bits), terms hh hm '
                                               'mh hl mm lh in 3 x v_mfma_f32_16x16x32_bf16: direct k3 stride-1 64->64 @16^3 (4), 32->32 @16^3 (2), 64->64 @8^3 (5), '
                                               'Conv3DTranspose stride 2 64->64 / 64->32 (2 + 2) (conv_split.hip) and 32->16 (2, conv_tr2m_bf16.hip).  Exact fp32 MFMA: the '
                                               'stride-2 forward layers, the 4^3 grids, first and last layer.  PCC_NO_F16S=1: (a) -> bf16 x 3 / exact fp32; PCC_NO_SPLIT=1: '
                                               'exact fp32 MFMA everywhere') if split and args.precision == 'fp32' else None),
                       'executed_flops_note': ('executed = fp32-equivalent multiply-adds the kernels perform (Winograd layers 16/36 of the taps, direct layers all); with '
                                               'mfma_operand_split most of them are issued as bf16 MFMAs, so the _executed fraction below is a work figure relative to the fp32 '
                                               'pipe, not the utilisation of one pipe') if split and args.precision == 'fp32' else None,
                       'bytes_per_block': n_bytes / n_blocks, 'decoded_points_per_block': n_pts / n_blocks,
                       'conv_tflops_whole_step_algorithmic': value * FLOPS_PER_BLOCK / 1e12,
                       # direct-convolution flops / time / peak: the Winograd layers execute 2.1-2.2x fewer multiplies
                       # than that, so this figure MAY EXCEED 1; the executed fraction next to it cannot
                       'conv_frac_of_fp32_mfma_peak_whole_step_algorithmic_winograd_may_exceed_1': value * FLOPS_PER_BLOCK / 1e12 / (PEAK_FP32_MFMA * world),
                       'conv_frac_of_fp32_mfma_peak_whole_step_executed': (value * exec_step / 1e12 / (PEAK_FP32_MFMA * world)) if args.precision == 'fp32' else None,
                       'executed_flops_per_block': exec_step if args.precision == 'fp32' else None,
                       'steady_state_ms_per_step': steady_ms,
                       'steady_state_blocks_per_s_per_gpu': (1e3 * BATCH / steady_ms) if steady_ms else None,
                       'steady_state_note': 'window from the arrival of chunk 2 to the arrival of chunk K-3 (3-deep pipeline filled, not draining); '
                                            '`value` / `ms_per_step` above include fill and drain'},
            'roofline': ({'bound': 'hbm', 'kernel': 'conv_f16_kernel<16> (fp16 storage, v_mfma_f32_16x16x32_f16): Conv3DTranspose 16->16 k3 s1 + fp16 residual, synthesis layer 8',
                          'achieved': alg_bytes16 / (avg_ms * 1e-3) / 1e9, 'peak': 8000.0, 'unit': 'GB/s', 'frac': alg_bytes16 / (avg_ms * 1e-3) / 1e9 / 8000.0,
                          'traffic': None, 'algorithmic_bytes_per_launch': alg_bytes16, 'avg_launch_ms': avg_ms, 'launches_timed': len(kern_ms),
                          'note': 'fp16 mode: the layer is HBM-bound (fp16 MFMA is 16x the fp32 rate); achieved = fp16 bytes of input + residual + '
                                  'output / HIP-event launch time; HBM3E peak 8 TB/s, ~6.3 TB/s achievable (MI355X_MICROARCH.md)'}
                         if args.precision == 'fp16' else dom_roofline),
        }
        if step_profile:
            ms_step = 1e3 * elapsed / args.steps
            names = {L.PCC_NET_ANALYSIS_PROGRESSIVE_V2: 'analysis', L.PCC_NET_HYPER_ANALYSIS: 'hyper_analysis',
                     L.PCC_NET_HYPER_SYNTHESIS: 'hyper_synthesis', L.PCC_NET_SYNTHESIS_PROGRESSIVE_V2: 'synthesis'}
            layers, tot_us, tot_bytes, pipe_s = [], 0.0, 0.0, 0.0
            for r in step_profile:
                if not r['launch_ms']:
                    continue
                med = float(np.median(r['launch_ms']))
                roofs, nbytes, pflops, peak = layer_roofs(ctx, r, args.chunk, med)
                tot_us += 1e3 * med * r['calls']; tot_bytes += nbytes * r['calls']; pipe_s += pflops / (peak * 1e12) * r['calls']
                layers.append(dict(layer=f"{names[r['transform']]}/{r['layer']}", shape=f"{r['cin']}->{r['cout']} k{r['k']} s{r['stride']}{' T' if r['transposed'] else ''} @{r['d_in']}^3"
                                   + (' +res' if r['residual'] else ''), launches=r['calls'], median_us=1e3 * med, share_of_step=1e3 * med * r['calls'] / (1e3 * ms_step), **roofs))
            layers.sort(key=lambda e: -e['share_of_step'])
            dom_roofline['step'] = {
                'ms_per_step': ms_step, 'conv_layers_sum_ms': tot_us / 1e3, 'conv_launches_per_step': sum(r['calls'] for r in step_profile),
                'residual_ms': ms_step - tot_us / 1e3,
                'residual_is': 'ms_per_step - sum(median launch time x launches): element-wise / packing / threshold kernels, queue gaps and host shadow, '
                               'MINUS whatever the encoder and decoder streams overlap (can be negative)',
                'list_covers_frac_of_step': tot_us / 1e3 / ms_step,
                'unfused_algorithmic_bytes_per_step': tot_bytes, 'hbm_frac_whole_step': tot_bytes / (ms_step * 1e-3) / 1e9 / PEAK_HBM,
                'time_if_every_layer_ran_at_6300_gbs_ms': tot_bytes / 6.3e12 * 1e3,
                'time_if_every_multiply_ran_at_its_pipe_peak_ms': pipe_s * 1e3, 'pipe_frac_whole_step': pipe_s * 1e3 / ms_step,
                'how': 'HIP events around one layer per pass (3 steps each, after the timed region, pipeline running); kernel_family from pcc_conv_kernel_family',
                'layers': layers}
        out['ab_exact_fp32'] = ab_exact
        out['secondary'] = secondary
        out['multi_gpu'] = multi
        if world == 1 and not args.no_cpu_baseline and args.workload == 'configs1':
            out['cpu_baseline'] = cpu_baseline(model, w, x[:4].cpu().numpy())
        else:
            out['cpu_baseline'] = None
        # anything a native library still holds in C stdio buffers (RCCL's version banner) goes out first: the JSON line is the last line
        import ctypes
        ctypes.CDLL(None).fflush(None)
        sys.stdout.flush()
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    sys.exit(main() or 0)
