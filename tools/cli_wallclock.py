"""End-to-end wall clock of the drop-in itself: compress_octree -> decompress_octree on the configs[2] stand-in
cloud (or PCC_BENCH_CLOUD=<vox10 .ply>), as the two separate processes a user of the reference starts, broken into phases by the
opt-in stamps of pcc_geo_cnn_v2_amd/utils/cli_timing.py.  Each CLI is run twice: `cold` = the first process on the box (page cache,
code objects and the GPU's clocks cold), `warm` = the same command again.

    python tools/cli_wallclock.py [--d2]        -> one markdown table on stdout (profiles/r06_cli_wallclock.md)
"""
import json, os, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np


def run(mod, argv, tag, tmp):
    tj = os.path.join(tmp, f'{tag}.json')
    env = dict(os.environ, PCC_CLI_TIMING_JSON=tj, PYTHONPATH=ROOT)
    t0 = time.time()
    r = subprocess.run([sys.executable, '-m', f'pcc_geo_cnn_v2_amd.{mod}'] + argv, env=env, cwd=ROOT, capture_output=True, text=True)
    t1 = time.time()
    assert r.returncode == 0, r.stderr[-2000:]
    marks = json.load(open(tj))['marks']
    out, prev = [], t0
    for name, t in marks:
        out.append((name, t - prev)); prev = t
    out.append(('exit (teardown: context, threads, interpreter)', t1 - prev))
    return out, t1 - t0


def main():
    import bench
    from pcc_geo_cnn_v2_amd.utils import pc_io
    tmp = tempfile.mkdtemp(prefix='pcc_cli_')
    cloud = os.environ.get('PCC_BENCH_CLOUD')
    if not cloud:
        pts = bench.standin_cloud()
        cloud = os.path.join(tmp, 'standin_vox10.ply')
        pc_io.write_df(cloud, pc_io.pa_to_df(pts))
    ck = os.path.join(tmp, 'ck')
    subprocess.run([sys.executable, '-m', 'pcc_geo_cnn_v2_amd.init_checkpoint', '--model_config', 'c3p', '--checkpoint_dir', ck], cwd=ROOT, check=True,
                   env=dict(os.environ, PYTHONPATH=ROOT), capture_output=True)
    common = ['--checkpoint_dir', ck, '--model_config', 'c3p']
    enc = ['--input_files', cloud, '--output_files', os.path.join(tmp, 'a.bin'), '--resolution', '1024', '--octree_level', '4', '--opt_metrics', 'd1_mse'] + common
    if '--fixed' in sys.argv:
        enc.append('--fixed_threshold')
    dec = ['--input_files', os.path.join(tmp, 'a.bin'), '--output_files', os.path.join(tmp, 'a.dec.ply')] + common
    rows = {}
    for state in ('cold', 'warm'):
        rows[('compress_octree', state)] = run('compress_octree', enc, f'enc_{state}', tmp)
        rows[('decompress_octree', state)] = run('decompress_octree', dec, f'dec_{state}', tmp)
    n_pts = len(pc_io.load_pc(cloud))
    print(f'# CLI wall clock: `compress_octree` -> `decompress_octree`, c3p, {os.path.basename(cloud)} ({n_pts} points, resolution 1024, octree level 4, '
          f'{"fixed threshold" if "--fixed" in sys.argv else "adaptive threshold search on d1_mse (the CLI default metric)"}), one MI355X, separate processes\n')
    print(f'container: {os.path.getsize(os.path.join(tmp, "a.bin"))} bytes; host cores usable: {len(os.sched_getaffinity(0))}\n')
    for mod in ('compress_octree', 'decompress_octree'):
        cold, tc = rows[(mod, 'cold')]
        warm, tw = rows[(mod, 'warm')]
        print(f'## {mod}: {tc:.2f} s cold, {tw:.2f} s warm\n\n| phase | cold s | warm s | share of warm |\n|---|---|---|---|')
        for (name, c), (_, w) in zip(cold, warm):
            print(f'| {name} | {c:.3f} | {w:.3f} | {100 * w / tw:.1f} % |')
        print()


if __name__ == '__main__':
    main()
