"""Worker for tests/test_sharding_cpu.py: one of N gloo ranks on CPU."""
import os
import pickle
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def fake_encode_block_range(self, sess, blocks, resolution, with_normals=False, opt_metrics=('d1_mse',),
                            max_deltas=(np.inf,), fixed_threshold=False, debug=False):
    """Deterministic stand-in for the GPU work of one shard: strings derived from the block content."""
    strings, thr, pts = [], [], []
    for b in blocks:
        key = int(np.asarray(b)[:, :3].sum()) % 251
        strings.append((bytes([key]) * (key % 7 + 1), bytes([255 - key])))
        thr.append([128])
        pts.append([np.asarray(b)[:, :3].astype(np.float32)])
    return strings, thr, pts, ['d1_mse_inf'], [None] * len(blocks)


def fake_encode_two_metrics(self, sess, blocks, resolution, with_normals=False, opt_metrics=('d1_mse',),
                            max_deltas=(np.inf,), fixed_threshold=False, debug=False):
    """Two candidates per block: the block itself (perturbed) for d1, a thinned copy for d2."""
    strings, thr, pts = [], [], []
    for b in blocks:
        p = np.asarray(b)[:, :3].astype(np.float32)
        key = int(p.sum()) % 251
        strings.append((bytes([key]) * (key % 5 + 1), bytes([key ^ 1]) * 2))
        thr.append([100 + key % 3, 140 + key % 5])
        pts.append([np.unique(np.clip(p + (key % 2), 0, 15), axis=0), p[::2].copy()])
    return strings, thr, pts, ['d1_mse_inf', 'd2_mse_inf'], [None] * len(blocks)


def fake_decompress(model, blocks):
    """decompress_blocks with the GPU part replaced: each 'compressed block' decodes to its own points."""
    from pcc_geo_cnn_v2_amd import sharding
    from pcc_geo_cnn_v2_amd.model_types import CompressionModel
    orig = CompressionModel.decompress_blocks

    def local(self, sess, blks, x_shape, debug=False):
        if sharding.world_info()[1] > 1 and not getattr(self, '_in_shard', False):
            return orig(self, sess, blks, x_shape, debug)
        return [np.asarray(b)[:, :3].astype(np.float32) for b in blks], [None] * len(blks)
    CompressionModel.decompress_blocks = local
    try:
        out = model.decompress_blocks(None, blocks, [16, 16, 16])
    finally:
        CompressionModel.decompress_blocks = orig
    return out[0]


def main():
    out_path = sys.argv[1]
    dist.init_process_group('gloo')
    rank, world = dist.get_rank(), dist.get_world_size()
    from pcc_geo_cnn_v2_amd import sharding
    from pcc_geo_cnn_v2_amd.model_configs import ModelConfigType
    from pcc_geo_cnn_v2_amd.model_types import CompressionModel
    from pcc_geo_cnn_v2_amd.utils.octree_coding import partition_octree
    res = {}
    calls = []
    for name in ('all_gather', 'gather', 'all_reduce', 'broadcast', 'all_gather_object', 'gather_object'):
        def counted(*a, _f=getattr(dist, name), _n=name, **k):
            calls.append(_n)
            return _f(*a, **k)
        setattr(dist, name, counted)
    # 1. typed collectives: ragged rows to every rank, ragged bytes / rows to rank 0
    res['rows'] = sharding.all_gather_rows(np.arange((rank + 1) * 2, dtype=np.int64).reshape(rank + 1, 2) + 100 * rank)
    res['bytes'] = sharding.gather_bytes(bytes(range(rank * 3 + 1)))
    res['frows'] = sharding.gather_rows(np.full((2 - rank, 3), rank + 0.5, np.float32))
    res['ranges'] = [sharding.shard_range(n, rank, world) for n in (0, 1, 5, 8, 13)]
    # 2. compress_blocks assembly over shards == single process
    rng = np.random.default_rng(0)
    pts = np.unique(rng.integers(0, 64, (400, 3)), axis=0).astype(np.float64)
    blocks, binstr = partition_octree(pts, [0, 0, 0], [64] * 3, 2)
    CompressionModel.encode_block_range = fake_encode_block_range
    model = ModelConfigType['c3p'].build()
    del calls[:]
    data_list, metadata, _ = model.compress_blocks(None, blocks, binstr, pts, 64, 2, fixed_threshold=True)
    res['calls_compress'] = list(calls)
    res['data_list'] = data_list
    res['metrics'] = metadata[0]['metrics']
    res['full'] = metadata[0].get('blocks_full')
    res['n_blocks'] = len(blocks)
    # 3. with normals and two optimisation targets (d1 + d2 groups): the candidate lists differ per metric
    nrm = rng.standard_normal((len(pts), 3))
    pn = np.hstack([pts, nrm / np.linalg.norm(nrm, axis=1, keepdims=True)])
    blocks_n, binstr_n = partition_octree(pn, [0, 0, 0], [64] * 3, 2)
    CompressionModel.encode_block_range = fake_encode_two_metrics
    del calls[:]
    dl, md, _ = model.compress_blocks(None, blocks_n, binstr_n, pn, 64, 2, with_normals=True, opt_metrics=['d1_mse', 'd2_mse'],
                                      need_points=False)
    res['calls_two'] = list(calls)
    res['two'] = dict(data_list=dl, idx=[m['idx'] for m in md], metrics=[m['metrics'] for m in md],
                      has_points=['blocks_full' in m for m in md])
    # 3b. the same cloud when the MIN keys are "too many" to ride in the all_gather: the three-collective path (all_reduce(MIN) kept)
    os.environ['PCC_KEY_GATHER_MAX_BYTES'] = '0'
    del calls[:]
    dl3, md3, _ = model.compress_blocks(None, blocks_n, binstr_n, pn, 64, 2, with_normals=True, opt_metrics=['d1_mse', 'd2_mse'],
                                        need_points=False)
    del os.environ['PCC_KEY_GATHER_MAX_BYTES']
    res['calls_two_big'] = list(calls)
    res['two_big'] = dict(data_list=dl3, idx=[m['idx'] for m in md3], metrics=[m['metrics'] for m in md3])
    # 4. decompress_blocks: decoded points to rank 0
    CompressionModel._in_shard = False
    model.decompress_local = True
    del calls[:]
    res['dec'] = fake_decompress(model, blocks)
    res['calls_dec'] = list(calls)
    with open(f'{out_path}.{rank}', 'wb') as f:
        pickle.dump(res, f)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == '__main__':
    main()
