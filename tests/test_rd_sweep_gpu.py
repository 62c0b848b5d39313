"""GPU: BASELINE.json configs[3] -- a rate-distortion sweep over >= 4 weight sets of the c3p graph through the real CLIs
(compress_octree -> decompress_octree -> ev_report), tools/rd_sweep.py."""
import csv
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_rd_sweep_designed_rate_points(tmp_path):
    sys.path.insert(0, os.path.join(ROOT, 'tools'))
    import rd_sweep
    out = str(tmp_path / 'rd')
    rows = rd_sweep.main(['--out', out, '--resolution', '256', '--octree_level', '2', '--levels', '1', '3', '4', '5', '6',
                          '--batch_size', '16'])
    # the asserts inside: enc/dec bit-consistent, decoded set == closed form, bpp and D1-PSNR strictly increasing
    assert len(rows) == 5
    with open(os.path.join(out, 'rd.csv')) as f:
        tab = list(csv.reader(f))
    assert tab[0] == ['eval_id', 'label', 'metric', 'mode_id', 'opt_group', 'pc_name', 'x', 'y', 'ylabel']   # data.csv columns
    assert len(tab) == 6 and all(r[1] == 'c4' and r[2] == 'd1_psnr' for r in tab[1:])
    x = [float(r[6]) for r in tab[1:]]
    y = [float(r[7]) for r in tab[1:]]
    assert x == sorted(x) and y == sorted(y) and y[-1] - y[0] > 5.0
