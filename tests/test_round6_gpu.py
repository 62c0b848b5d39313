"""Round 6: the configs[2] bench workload (one cloud, sharded, collectives inside the clock) and the fp16-split side channel."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('force_dist', [False, True])
def test_bench_configs2_runs_one_cloud_through_the_sharded_entry_points(force_dist):
    """`bench.py --workload configs2` (BASELINE.json configs[2]; /root/reference/src/compress_octree.py:97-105, decompress_octree.py:60-66):
    a labelled strong-scaling line; with a one-rank RCCL group the closing collectives really run and are accounted for."""
    env = dict(os.environ, PYTHONPATH=ROOT)
    if force_dist:
        env.update(PCC_BENCH_FORCE_DIST='1', MASTER_ADDR='127.0.0.1', MASTER_PORT='29631')
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--workload', 'configs2', '--steps', '2', '--warmup', '2'], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    d = json.loads(r.stdout.strip().splitlines()[-1])
    assert d['scaling'] == 'strong' and d['n_gpus'] == 1 and 'configs2' in d['metric'] and d['value'] > 0
    assert 150 < d['config']['blocks'] < 2000 and d['config']['decoded_points'] > 0 and d['config']['container_bytes_gzip'] > 0
    # a one-rank group takes the unsharded path (sharding.world_info() == (0, 1)): no collective is issued, and the line says so
    assert d['final_collectives_calls_per_step'] == 0 and d['final_collectives_ms'] == 0
    assert set(d['config']['phase_ms_per_step_rank0']) == {'encode', 'handover', 'decode'}
