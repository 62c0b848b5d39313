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
    # typed collectives
    for r in range(2):
        assert two[r]['rows'].tolist() == [[0, 1], [100, 101], [102, 103]]
    assert two[0]['bytes'] == [bytes(range(1)), bytes(range(4))] and two[1]['bytes'] is None
    assert two[0]['frows'].tolist() == [[0.5] * 3, [0.5] * 3, [1.5] * 3] and two[1]['frows'] is None
    assert two[0]['ranges'] == [(0, 0), (0, 1), (0, 3), (0, 4), (0, 7)]
    assert two[1]['ranges'] == [(0, 0), (1, 1), (3, 5), (4, 8), (7, 13)]
    assert one[0]['n_blocks'] > 4
    # rank 0 assembles exactly the single-process result (same strings, same thresholds, same block order, same points)
    assert two[0]['data_list'] == one[0]['data_list'] and two[1]['data_list'] is None
    assert np.array_equal(two[0]['full'], one[0]['full']) and two[1]['full'] is None
    for r in range(2):   # the metrics come from all_reduce'd partial sums: every rank holds them, D1 exactly
        for k, v in one[0]['metrics'].items():
            assert two[r]['metrics'][k] == v, k
    # normals + two optimisation groups, no point gather
    assert two[0]['two']['data_list'] == one[0]['two']['data_list'] and two[1]['two']['data_list'] is None
    for r in range(2):
        assert two[r]['two']['idx'] == one[0]['two']['idx'] == [0, 1]
        assert two[r]['two']['has_points'] == [False, False]
        for ma, mb in zip(two[r]['two']['metrics'], one[0]['two']['metrics']):
            for k, v in mb.items():
                if k.startswith('d1'):
                    assert ma[k] == v, k
                else:   # D2 depends on which of several equidistant neighbours is taken (cross-shard ties: lowest rank)
                    assert np.isclose(ma[k], v, rtol=0.05), (k, ma[k], v)
    # the number of collectives DESIGN_HISTORY.md §6 states: no size exchanges (shard sizes are a function of (n_blocks, world))
    for r in range(2):
        # round 5: TWO collectives per cloud, whatever the number of candidates (SURVEY.md 8e's "single gather" +
        # the strings): the MIN keys ride in the row all_gather, the partial tallies with the strings; when the keys are too many to move
        # `world` times (8 B per input point and candidate; bound PCC_KEY_GATHER_MAX_BYTES, default 64 MB) the MIN stays an all_reduce: three
        assert two[r]['calls_compress'] == ['all_gather', 'all_gather', 'gather']             # last: --dec_files points
        assert two[r]['calls_two'] == ['all_gather', 'all_gather']                           # 2 candidates, need_points=False
        assert two[r]['calls_two_big'] == ['all_reduce', 'all_gather', 'gather']
        assert two[r]['two_big']['idx'] == two[r]['two']['idx']
        assert two[r]['two_big']['metrics'] == two[r]['two']['metrics']                        # same keys, same MIN, same tallies: same doubles
    assert two[0]['two_big']['data_list'] == two[0]['two']['data_list'] and two[1]['two_big']['data_list'] is None
    for r in range(2):
        assert two[r]['calls_dec'] == ['all_gather', 'gather']
    assert one[0]['calls_compress'] == one[0]['calls_two'] == one[0]['calls_two_big'] == one[0]['calls_dec'] == []
    # decoder: all points on rank 0, in block order
    assert two[1]['dec'] is None and len(two[0]['dec']) == one[0]['n_blocks']
    for a, b in zip(two[0]['dec'], one[0]['dec']):
        assert np.array_equal(a, b)


def test_bench_self_launches_one_rank_per_gpu(tmp_path):
    """`python bench.py --gpus 2` with no launcher in the environment re-execs itself under torch.distributed.run and reports
    n_gpus = 2 with both ranks' blocks (dry run: gloo on CPU, launcher + collective skeleton only); and it refuses to run
    when fewer GPUs are visible than ranks instead of printing a mislabeled 1-GPU line."""
    import json
    root = os.path.dirname(HERE)
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
    env['OMP_NUM_THREADS'] = '1'
    p = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '2', '--steps', '3', '--warmup', '0', '--dry-run'],
                       env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out['n_gpus'] == 2 and out['dry_run'] is True and out['blocks_total'] == 2 * 3 * 32
    # no GPUs here: the real bench must fail loudly, not fall back to one rank
    p = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '2', '--steps', '1', '--warmup', '0'],
                       env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode != 0 and 'refusing to run a mislabeled bench' in p.stderr
    assert not [l for l in p.stdout.splitlines() if l.startswith('{')]
