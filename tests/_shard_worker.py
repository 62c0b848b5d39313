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


def main():
    out_path = sys.argv[1]
    dist.init_process_group('gloo')
    rank, world = dist.get_rank(), dist.get_world_size()
    from pcc_geo_cnn_v2_amd import sharding
    from pcc_geo_cnn_v2_amd.model_configs import ModelConfigType
    from pcc_geo_cnn_v2_amd.model_types import CompressionModel
    from pcc_geo_cnn_v2_amd.utils.octree_coding import partition_octree
    res = {}
    # 1. gather_objects: ragged payloads, every rank gets the rank-ordered list
    got = sharding.gather_objects({'rank': rank, 'blob': bytes(range(rank * 3 + 1))})
    res['gather'] = got
    res['ranges'] = [sharding.shard_range(n, rank, world) for n in (0, 1, 5, 8, 13)]
    # 2. compress_blocks assembly over shards == single process
    rng = np.random.default_rng(0)
    pts = np.unique(rng.integers(0, 64, (400, 3)), axis=0).astype(np.float64)
    blocks, binstr = partition_octree(pts, [0, 0, 0], [64] * 3, 2)
    CompressionModel.encode_block_range = fake_encode_block_range
    model = ModelConfigType['c3p'].build()
    data_list, metadata, _ = model.compress_blocks(None, blocks, binstr, pts, 64, 2, fixed_threshold=True)
    res['data_list'] = data_list
    res['psnr'] = metadata[0]['metrics']['d1_psnr']
    res['n_blocks'] = len(blocks)
    with open(f'{out_path}.{rank}', 'wb') as f:
        pickle.dump(res, f)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == '__main__':
    main()
