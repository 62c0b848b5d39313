"""GPU: the compress_octree / decompress_octree CLIs end to end (PLY in -> .ply.bin -> PLY out)."""
import gzip
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from pcc_geo_cnn_v2_amd import model_syntax
from pcc_geo_cnn_v2_amd.utils import pc_io

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _cloud(res, seed):
    rng = np.random.default_rng(seed)
    g = np.stack(np.meshgrid(*[np.arange(res)] * 3, indexing='ij'), -1).reshape(-1, 3)
    d = np.linalg.norm(g - res / 2 + 0.3, axis=1)
    return g[np.abs(d - res * 0.37) < 0.7].astype(np.float32)


@pytest.mark.parametrize('cfg,fixed', [('c3p', True), ('c1', True), ('c3p', False)])
def test_cli_roundtrip(tmp_path, cfg, fixed):
    res, level = 128, 2   # 32^3 blocks
    pts = _cloud(res, 0)
    src = str(tmp_path / 'in.ply')
    pc_io.write_df(src, pc_io.pa_to_df(pts))
    ck = str(tmp_path / 'ckpt')
    env = dict(os.environ, PYTHONPATH=ROOT)
    run = lambda *a: subprocess.run([sys.executable, '-m'] + list(a), cwd=ROOT, env=env, check=True, capture_output=True, text=True)
    run('pcc_geo_cnn_v2_amd.init_checkpoint', '--model_config', cfg, '--checkpoint_dir', ck,
        *(['--final_bias', '0.47', '--gain_analysis', '2.2'] if cfg == 'c1' else []))
    out, dec_enc, dec = str(tmp_path / 'o' / 'in.ply.bin'), str(tmp_path / 'enc.ply'), str(tmp_path / 'dec.ply')
    args = ['pcc_geo_cnn_v2_amd.compress_octree', '--input_files', src, '--output_files', out, '--dec_files', dec_enc,
            '--checkpoint_dir', ck, '--model_config', cfg, '--resolution', str(res), '--octree_level', str(level),
            '--opt_metrics', 'd1_mse', '--debug', '--batch_size', '5']
    run(*(args + (['--fixed_threshold'] if fixed else [])))
    with gzip.open(out, 'rb') as f:
        r, l, binstr, blocks = model_syntax.load_compressed_file(f)
    assert (r, l) == (res, level) and len(blocks) > 8
    assert all(len(s) == (1 if cfg == 'c1' else 2) for s, _ in blocks)
    if fixed:
        assert all(t == 128 for _, t in blocks)
    met = json.load(open(out + '.enc.metric.json'))
    assert 'd1_psnr' in met
    # --debug: the decoder re-checks every intermediate against the encoder dumps (bit-exact, no retries)
    run('pcc_geo_cnn_v2_amd.decompress_octree', '--input_files', out, '--output_files', dec, '--checkpoint_dir', ck,
        '--model_config', cfg, '--debug', '--batch_size', '7')
    a, b = pc_io.load_pc(dec_enc), pc_io.load_pc(dec)
    assert a.shape == b.shape and np.array_equal(a, b)       # decoder output == encoder-side reconstruction
    assert len(b) > 0 and b.min() >= 0 and b.max() < res


def test_cli_missing_checkpoint_fails_like_reference(tmp_path):
    src = str(tmp_path / 'in.ply')
    pc_io.write_df(src, pc_io.pa_to_df(_cloud(64, 1)))
    p = subprocess.run([sys.executable, '-m', 'pcc_geo_cnn_v2_amd.compress_octree', '--input_files', src, '--output_files',
                        str(tmp_path / 'x.bin'), '--checkpoint_dir', str(tmp_path / 'none'), '--model_config', 'c3p',
                        '--resolution', '64', '--octree_level', '1', '--opt_metrics', 'd1_mse'], cwd=ROOT, env=dict(os.environ, PYTHONPATH=ROOT),
                       capture_output=True, text=True)
    assert p.returncode != 0 and 'was not found' in p.stderr


def test_cli_sharded_over_two_ranks_writes_the_single_process_file(tmp_path):
    """compress_octree / decompress_octree under torch.distributed.run with two ranks (both on this box's one GPU, gloo
    collectives): rank 0 writes byte-identical .ply.bin / .enc.metric.json / decoded .ply to the single-process run
    (contiguous Morton ranges per rank, one gather at the end)."""
    res, level = 128, 2   # 64 blocks of 32^3
    pts = _cloud(res, 3)
    src = str(tmp_path / 'in.ply')
    pc_io.write_df(src, pc_io.pa_to_df(pts))
    ck = str(tmp_path / 'ckpt')
    env = dict(os.environ, PYTHONPATH=ROOT)
    subprocess.run([sys.executable, '-m', 'pcc_geo_cnn_v2_amd.init_checkpoint', '--model_config', 'c3p', '--checkpoint_dir', ck],
                   cwd=ROOT, env=env, check=True, capture_output=True)

    def run(tag, nranks):
        out, dec_enc, dec = str(tmp_path / f'{tag}.ply.bin'), str(tmp_path / f'{tag}.enc.ply'), str(tmp_path / f'{tag}.dec.ply')
        launcher = [sys.executable, '-m'] if nranks == 1 else \
            [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={nranks}', '--master-addr', '127.0.0.1',
             '--master-port', '29533', '-m']
        e = dict(env, PCC_DIST_BACKEND='gloo', PCC_DIST_SAME_GPU='1')
        for args in (['pcc_geo_cnn_v2_amd.compress_octree', '--input_files', src, '--output_files', out, '--dec_files', dec_enc,
                      '--checkpoint_dir', ck, '--model_config', 'c3p', '--resolution', str(res), '--octree_level', str(level),
                      '--opt_metrics', 'd1_mse', '--batch_size', '7'],
                     ['pcc_geo_cnn_v2_amd.decompress_octree', '--input_files', out, '--output_files', dec, '--checkpoint_dir', ck,
                      '--model_config', 'c3p', '--batch_size', '5']):
            p = subprocess.run(launcher + args, cwd=ROOT, env=e, capture_output=True, text=True, timeout=600)
            assert p.returncode == 0, p.stderr[-3000:]
        return out, dec_enc, dec

    a, b = run('one', 1), run('two', 2)
    with gzip.open(a[0], 'rb') as f1, gzip.open(b[0], 'rb') as f2:
        assert f1.read() == f2.read()                               # the same container, byte for byte (adaptive thresholds included)
    assert json.load(open(a[0] + '.enc.metric.json')) == json.load(open(b[0] + '.enc.metric.json'))
    for i in (1, 2):
        assert np.array_equal(pc_io.load_pc(a[i]), pc_io.load_pc(b[i]))
    assert np.array_equal(pc_io.load_pc(b[1]), pc_io.load_pc(b[2]))   # decoder == encoder-side reconstruction


def test_cli_stream_carries_its_codec_numerics_and_the_decoder_refuses_another_family(tmp_path):
    """sigma-hat selects the entropy coder's rows, so a stream written under one kernel family (exact-fp32 MFMA vs split-bf16,
    fp32 vs fp16 mode) must not be decoded under another.  compress_octree records the context's numerics tag in the gzip member header
    (the container bytes stay the reference's, /root/reference/src/model_syntax.py:20-35); decompress_octree refuses a mismatch, decodes
    with --ignore_numerics_tag, and decodes an untagged (reference-written) file with a warning.  The PCC_* switches reach the context
    through the environment at creation, once."""
    res, level = 64, 1
    src, ck = str(tmp_path / 'in.ply'), str(tmp_path / 'ckpt')
    pc_io.write_df(src, pc_io.pa_to_df(_cloud(res, 3)))
    env = dict(os.environ, PYTHONPATH=ROOT)
    run = lambda e, *a: subprocess.run([sys.executable, '-m'] + list(a), cwd=ROOT, env=e, capture_output=True, text=True)
    assert run(env, 'pcc_geo_cnn_v2_amd.init_checkpoint', '--model_config', 'c3p', '--checkpoint_dir', ck).returncode == 0
    out, dec = str(tmp_path / 'a.ply.bin'), str(tmp_path / 'dec.ply')
    enc = ['pcc_geo_cnn_v2_amd.compress_octree', '--input_files', src, '--output_files', out, '--checkpoint_dir', ck, '--model_config', 'c3p',
           '--resolution', str(res), '--octree_level', str(level), '--opt_metrics', 'd1_mse', '--fixed_threshold']
    p = run(env, *enc)
    assert p.returncode == 0, p.stderr[-2000:]
    tag = model_syntax.read_gzip_tag(out)
    assert tag is not None and tag.startswith('pcc_geo_cnn_v2_amd/k') and tag.endswith('/sw0000/fp32'), tag
    decode = ['pcc_geo_cnn_v2_amd.decompress_octree', '--input_files', out, '--output_files', dec, '--checkpoint_dir', ck, '--model_config', 'c3p']
    assert run(env, *decode).returncode == 0
    ref_pts = pc_io.load_pc(dec)
    # another kernel family on the decoder (PCC_NO_SPLIT=1 -> numerics switch 0x1): refused, loudly
    other = dict(env, PCC_NO_SPLIT='1')
    p = run(other, *decode)
    assert p.returncode != 0 and 'codec numerics' in p.stderr and 'sw0001' in p.stderr, p.stderr[-2000:]
    p = run(env, *(decode + ['--precision', 'fp16']))
    assert p.returncode != 0 and 'codec numerics' in p.stderr
    assert run(other, *(decode + ['--ignore_numerics_tag'])).returncode in (0, 1)      # tries; may legitimately fail on a desynchronised stream
    # the same switch on BOTH sides: a consistent pair again, tagged as such
    out2 = str(tmp_path / 'b.ply.bin')
    p = run(other, *(enc[:4] + [out2] + enc[5:]))
    assert p.returncode == 0, p.stderr[-2000:]
    assert model_syntax.read_gzip_tag(out2).endswith('/sw0001/fp32')
    assert run(other, 'pcc_geo_cnn_v2_amd.decompress_octree', '--input_files', out2, '--output_files', dec, '--checkpoint_dir', ck, '--model_config', 'c3p').returncode == 0
    # a file as the reference writes it (plain gzip.open, no tag): decoded, with a warning
    with gzip.open(out, 'rb') as f:
        payload = f.read()
    plain = str(tmp_path / 'plain.ply.bin')
    with gzip.open(plain, 'wb') as f:
        f.write(payload)
    p = run(env, 'pcc_geo_cnn_v2_amd.decompress_octree', '--input_files', plain, '--output_files', dec, '--checkpoint_dir', ck, '--model_config', 'c3p')
    assert p.returncode == 0 and 'no codec-numerics tag' in p.stderr, p.stderr[-2000:]
    assert np.array_equal(pc_io.load_pc(dec), ref_pts)
