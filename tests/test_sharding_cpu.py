"""CPU, world_size 2, gloo: the multi-GPU path (block sharding + the single gather) is correct by
construction -- the sharded result equals the single-process result."""
import os
import pickle
import socket
import subprocess
import sys

import numpy as np

from pcc_geo_cnn_v2_amd import sharding

HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run(world, out):
    port = _free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), LOCAL_RANK=str(r), MASTER_ADDR='127.0.0.1',
                   MASTER_PORT=str(port), OMP_NUM_THREADS='1')
        procs.append(subprocess.Popen([sys.executable, os.path.join(HERE, '_shard_worker.py'), out], env=env))
    for p in procs:
        assert p.wait(timeout=600) == 0
    return [pickle.load(open(f'{out}.{r}', 'rb')) for r in range(world)]


def test_shard_range_is_a_partition():
    for n in (0, 1, 7, 8, 9, 100):
        for w in (1, 2, 3, 8):
            rs = [sharding.shard_range(n, r, w) for r in range(w)]
            assert rs[0][0] == 0 and rs[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(rs, rs[1:]))
            assert max(hi - lo for lo, hi in rs) - min(hi - lo for lo, hi in rs) <= 1


def test_two_rank_sharded_compress_equals_single_process(tmp_path):
    two = _run(2, str(tmp_path / 'w2'))
    one = _run(1, str(tmp_path / 'w1'))
    for r in range(2):
        assert [g['rank'] for g in two[r]['gather']] == [0, 1]
        assert two[r]['gather'][1]['blob'] == bytes(range(4))
    assert two[0]['ranges'] == [(0, 0), (0, 1), (0, 3), (0, 4), (0, 7)]
    assert two[1]['ranges'] == [(0, 0), (1, 1), (3, 5), (4, 8), (7, 13)]
    assert one[0]['n_blocks'] > 4
    for r in range(2):   # every rank holds the complete, block-ordered result
        assert two[r]['data_list'] == one[0]['data_list']
        assert np.isclose(two[r]['psnr'], one[0]['psnr'])
