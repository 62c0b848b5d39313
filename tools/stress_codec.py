"""Randomised end-to-end sweep on the GPU: compress_blocks -> decompress_blocks with different chunkings, and the streaming
round trip, over configs, block edges, batch sizes and data formats.  Checks the size-independent properties: decoder x_hat ==
encoder x_hat bit for bit, decoded points == encoder-side points == np.argwhere(x_hat > thr) in order, roundtrip_stream ==
the block API.   python tools/stress_codec.py [seed] [cases]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pcc_geo_cnn_v2_amd import ops
from pcc_geo_cnn_v2_amd.model_configs import ModelConfigType
from pcc_geo_cnn_v2_amd.utils.octree_coding import partition_octree

ctx = ops.Context(0)
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
ncases = int(sys.argv[2]) if len(sys.argv) > 2 else 12
thr = np.float32(np.linspace(0, 1, 256)[128])


def weights(model, gain):
    w = model.get_weights()
    r = np.random.default_rng(7)
    for k in list(w):
        if k.endswith('/kernel'):
            w[k] = (w[k] * gain).astype(np.float32)
        if k.endswith('/bias') and not k.startswith('entropy'):
            w[k] = r.normal(0, 0.05, w[k].shape).astype(np.float32)
    last = max(int(k.split('/')[1]) for k in w if k.startswith('synthesis/'))
    w[f'synthesis/{last}/bias'] = np.array([0.47], np.float32)
    return w


bad = 0
for case in range(ncases):
    name = str(rng.choice(['c1', 'c2', 'c3', 'c3p']))
    v2 = name in ('c3', 'c3p')
    res = int(rng.choice([16, 32, 64]))          # (the octree needs a power-of-two cloud edge)
    nb = int(rng.integers(1, 9))
    df = str(rng.choice(['channels_first', 'channels_last']))
    be, bd = int(rng.integers(1, 6)), int(rng.integers(1, 6))
    R = res * 2
    blocks8 = []
    for i in range(nb):
        c = rng.uniform(res * 0.3, res * 0.7, 3); rad = rng.uniform(res * 0.2, res * 0.35)
        g = np.stack(np.meshgrid(*[np.arange(res)] * 3, indexing='ij'), -1).reshape(-1, 3)
        pts = g[np.abs(np.linalg.norm(g - c, axis=1) - rad) < 0.6]
        blocks8.append(np.unique(np.vstack([pts, rng.integers(0, res, (5, 3))]), axis=0).astype(np.float64))
    pts = np.vstack([b + np.array([(i & 1), (i >> 1) & 1, (i >> 2) & 1]) * res for i, b in enumerate(blocks8)])
    blocks, binstr = partition_octree(pts, [0, 0, 0], [R] * 3, 1)
    enc = ModelConfigType[name].build(batch_size=be, data_format=df)
    enc.compress([1, 1, res, res, res] if df == 'channels_first' else [1, res, res, res, 1])
    enc.set_weights(weights(enc, 2.2))
    data_list, metadata, dbg_e = enc.compress_blocks(ctx, blocks, binstr, pts, R, 1, fixed_threshold=True, debug=True)
    dec = ModelConfigType[name].build(batch_size=bd, data_format=df)
    dec.decompress()
    dec.set_weights({k: v for k, v in enc.get_weights().items() if not k.startswith(('analysis/', 'hyper_analysis/'))})
    dec_blocks, dbg_d = dec.decompress_blocks(ctx, data_list[0], [res] * 3, debug=True)
    ok = len(dec_blocks) == len(blocks)
    for j in range(len(blocks)):
        xh = dbg_d[j]['x_hat'][0, ..., 0]
        ok &= np.array_equal(dbg_e[j]['x_hat'], dbg_d[j]['x_hat']) and np.array_equal(metadata[0]['x_hat_list'][j], dec_blocks[j])
        ok &= np.array_equal(np.argwhere(xh > thr).astype(np.float32), dec_blocks[j])
    # the streaming round trip (the unit bench.py times) on the same blocks, chunked by `be`
    dense = enc._voxelize(enc._ctx(ctx), blocks, (res, res, res))
    chunks = [dense[i:i + be].contiguous() for i in range(0, len(blocks), be)]
    got = list(enc.roundtrip_stream(ctx, iter(chunks)))
    k = 0
    for strings, cnt_e, ptsl in got:
        for b in range(len(strings)):
            ok &= tuple(strings[b]) == tuple(data_list[0][k][0]) and int(cnt_e[b]) == len(dec_blocks[k]) and np.array_equal(ptsl[b], dec_blocks[k])
            k += 1
    ok &= k == len(blocks)
    print(('ok  ' if ok else 'FAIL'), dict(config=name, res=res, blocks=len(blocks), data_format=df, enc_batch=be, dec_batch=bd,
                                          points=int(sum(len(b) for b in dec_blocks))))
    bad += not ok
print(f'{ncases} cases, {bad} failures')
sys.exit(1 if bad else 0)
