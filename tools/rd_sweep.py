#!/usr/bin/env python
"""Rate-distortion sweep on ONE MI355X -- BASELINE.json configs[3] ("c4 (hyperprior), six lambda values, RD curve").

The reference sweeps the lambda list of /root/reference/src/ev_experiment.yml:25-29 (id c3p-a0.75, paper label c4; the c3p
graph with one trained checkpoint per lambda) through compress_octree.py -> decompress_octree.py -> pc_error and plots D1-PSNR
over bpp (ev_experiment.py:139-162, data.csv:243-246).  Trained checkpoints are not available offline, so the rate points
here are the six designed weight sets of init_checkpoint.make_cell_codec_weights (levels 1..6: "occupied cell" codecs with
cells from 4x8x8 down to 2^3 voxels) -- same graph, same CLIs, same container, same report; a directory of real checkpoints
(--checkpoint_dirs, one per lambda) drops into the same loop.

For every rate point: init_checkpoint -> compress_octree (fixed threshold, --dec_files) -> decompress_octree -> ev_report.
Checked: decoder output == encoder-side reconstruction (bit-consistent enc/dec), the decoded set == the closed-form
prediction for designed weights, bpp and D1-PSNR strictly increasing along the sweep.  Output: <out>/rd.csv in the column
order of the reference's data.csv (eval_id,label,metric,mode_id,opt_group,pc_name,x,y,ylabel) + rd.json.

    python tools/rd_sweep.py --out gpurun_out/rd [--resolution 512 --octree_level 3] [--input_pc cloud.ply]
"""
import argparse
import csv
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def synthetic_cloud(resolution, seed=0):
    """Voxelised torus-like shell filling a `resolution`^3 grid (stand-in for longdress_vox10_1300)."""
    rng = np.random.default_rng(seed)
    n = int(3.2 * resolution ** 2)
    u, v = rng.uniform(0, 2 * np.pi, n), rng.uniform(0, 2 * np.pi, n)
    R, r = 0.30 * resolution, 0.13 * resolution
    p = np.stack([(R + r * np.cos(v)) * np.cos(u), (R + r * np.cos(v)) * np.sin(u), r * np.sin(v) * 1.4], 1) + resolution / 2
    return np.unique(np.clip(np.round(p), 0, resolution - 1).astype(np.int64), axis=0).astype(np.float32)


def run_point(tag, ckpt, src, out_dir, resolution, level, batch_size):
    from pcc_geo_cnn_v2_amd import compress_octree, decompress_octree, ev_report
    from pcc_geo_cnn_v2_amd.utils import pc_io
    enc, dec_enc, dec = (os.path.join(out_dir, f'{tag}.ply.bin'), os.path.join(out_dir, f'{tag}.enc.ply'),
                         os.path.join(out_dir, f'{tag}.dec.ply'))
    compress_octree.compress(compress_octree.build_parser().parse_args(
        ['--input_files', src, '--output_files', enc, '--dec_files', dec_enc, '--checkpoint_dir', ckpt, '--model_config', 'c3p',
         '--resolution', str(resolution), '--octree_level', str(level), '--opt_metrics', 'd1_mse', '--fixed_threshold',
         '--batch_size', str(batch_size)]))
    decompress_octree.decompress(decompress_octree.build_parser().parse_args(
        ['--input_files', enc, '--output_files', dec, '--checkpoint_dir', ckpt, '--model_config', 'c3p', '--batch_size', str(batch_size)]))
    a, b = pc_io.load_pc(dec_enc), pc_io.load_pc(dec)
    assert a.shape == b.shape and np.array_equal(a, b), f'{tag}: decoder output differs from the encoder-side reconstruction'
    rep = ev_report.build_report(src, dec, enc, resolution)
    met = json.load(open(enc + '.enc.metric.json'))
    assert abs(met['d1_psnr'] - rep['d1_psnr']) < 0.01            # the reference's own check, ev_experiment.py:158-162
    return rep, b


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument('--out', required=True)
    ap.add_argument('--resolution', type=int, default=512)
    ap.add_argument('--octree_level', type=int, default=3, help='block edge = resolution / 2^level (64 for the defaults)')
    ap.add_argument('--levels', type=int, nargs='+', default=[1, 2, 3, 4, 5, 6], help='cell-codec rate points (designed weights)')
    ap.add_argument('--checkpoint_dirs', nargs='*', default=None, help='real checkpoints instead (one directory per lambda)')
    ap.add_argument('--input_pc', default=None, help='a real cloud (.ply) instead of the synthetic shell')
    ap.add_argument('--pc_name', default=None)
    ap.add_argument('--batch_size', type=int, default=32)
    ap.add_argument('--keep', action='store_true', help='keep the per-point .ply / .ply.bin / checkpoint files (tens of MB each)')
    a = ap.parse_args(argv)
    from pcc_geo_cnn_v2_amd.init_checkpoint import cell_codec_expected_points, cell_shape, make_cell_codec_weights
    from pcc_geo_cnn_v2_amd.utils import pc_io
    os.makedirs(a.out, exist_ok=True)
    if a.input_pc:
        src, pts = a.input_pc, pc_io.load_pc(a.input_pc)
    else:
        pts = synthetic_cloud(a.resolution)
        src = os.path.join(a.out, 'input.ply')
        pc_io.write_df(src, pc_io.pa_to_df(pts))
    pc_name = a.pc_name or os.path.splitext(os.path.basename(src))[0]
    rows = []
    if a.checkpoint_dirs:
        points = [(os.path.basename(os.path.normpath(d)), d, None) for d in a.checkpoint_dirs]
    else:
        points = []
        for lv in a.levels:
            ck = os.path.join(a.out, f'ckpt_cells{lv}')
            os.makedirs(ck, exist_ok=True)
            np.savez(os.path.join(ck, 'model.npz'), **make_cell_codec_weights(lv))
            points.append((f'cells{lv}', ck, lv))
    for tag, ck, lv in points:
        rep, decoded = run_point(tag, ck, src, a.out, a.resolution, a.octree_level, a.batch_size)
        if lv is not None:
            exp = cell_codec_expected_points(pts, lv)
            got = decoded.astype(np.int64)
            got = got[np.lexsort((got[:, 2], got[:, 1], got[:, 0]))]
            assert np.array_equal(got, exp), f'{tag}: decoded set differs from the closed-form prediction'
            rep['cell_shape'] = list(cell_shape(lv))
        rep['mode'] = tag
        rows.append(rep)
        if not a.keep:
            for fn in (f'{tag}.ply.bin', f'{tag}.enc.ply', f'{tag}.dec.ply', f'{tag}.ply.bin.enc.metric.json'):
                os.remove(os.path.join(a.out, fn))
            if lv is not None:
                os.remove(os.path.join(ck, 'model.npz'))
                os.rmdir(ck)
        print(f"{tag:10s} bpp {rep['pos_bits_per_input_point']:.4f}  D1-PSNR {rep['d1_psnr']:.3f} dB  "
              f"{rep['pos_total_size_in_bytes']} bytes, {len(decoded)} decoded points", flush=True)
    if not a.checkpoint_dirs:
        bpp = [r['pos_bits_per_input_point'] for r in rows]
        psnr = [r['d1_psnr'] for r in rows]
        assert all(x < y for x, y in zip(bpp, bpp[1:])), f'bpp not increasing along the sweep: {bpp}'
        assert all(x < y for x, y in zip(psnr, psnr[1:])), f'D1-PSNR not increasing along the sweep: {psnr}'
    with open(os.path.join(a.out, 'rd.csv'), 'w', newline='') as f:
        wr = csv.writer(f)
        wr.writerow(['eval_id', 'label', 'metric', 'mode_id', 'opt_group', 'pc_name', 'x', 'y', 'ylabel'])     # data.csv:1
        for r in rows:
            wr.writerow(['main', 'c4', 'd1_psnr', 'c3p-a0.75' if a.checkpoint_dirs else f"c3p-designed-{r['mode']}", 'd1', pc_name,
                         r['pos_bits_per_input_point'], r['d1_psnr'], 'D1 PSNR (dB)'])
    with open(os.path.join(a.out, 'rd.json'), 'w') as f:
        json.dump(dict(pc_name=pc_name, input_points=len(pts), resolution=a.resolution, octree_level=a.octree_level, points=rows), f, indent=2)
    if not a.keep and not a.input_pc:
        os.remove(src)
    return rows


if __name__ == '__main__':
    main()
