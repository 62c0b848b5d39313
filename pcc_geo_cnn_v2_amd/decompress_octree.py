"""Decoder CLI -- same flags and files as /root/reference/src/decompress_octree.py:148-182.

  python -m pcc_geo_cnn_v2_amd.decompress_octree --input_files a.ply.bin --output_files a.dec.ply \\
         --checkpoint_dir models/c3p/1.00e-04 --model_config c3p [--debug]

`--debug` reloads the encoder-side dumps (`.enc.data.npz`, `.enc.blocks/`) and checks every intermediate
and the decoded blocks for exact equality.  The reference needs up to 100 retries there because its GPU
results are not reproducible (decompress_octree.py:69-131); this implementation is bit-deterministic, so a
mismatch is an error, not a retry.
"""
import argparse
import gzip
import logging
import os

import numpy as np

logging.basicConfig(level=logging.INFO,
                    format='%(asctime)s.%(msecs)03d %(levelname)s %(module)s - %(funcName)s: %(message)s',
                    datefmt='%Y-%m-%d %H:%M:%S')
logger = logging.getLogger(__name__)


def read_pcs(length, folder):
    from .utils import pc_io
    return [pc_io.load_pc(os.path.join(folder, f'{j}.ply')) for j in range(length)]


def decompress(args):
    from .utils import cli_timing as T
    T.mark('process_start_to_main')
    import torch
    from . import ops, sharding
    from .model_configs import ModelConfigType
    from .model_syntax import check_numerics_tag, load_compressed_file, read_gzip_tag
    from .utils import pc_io
    from .utils.octree_coding import departition_octree
    T.mark('imports')

    assert args.data_format in ['channels_first', 'channels_last']
    assert len(args.input_files) == len(args.output_files)
    assert args.model_config in ModelConfigType.keys()

    world = int(os.environ.get('WORLD_SIZE', '1'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world > 1:
        import torch.distributed as dist
        # PCC_DIST_BACKEND=gloo + PCC_DIST_SAME_GPU=1: every rank on GPU 0 with host-side collectives -- lets a 1-GPU box run the
        # sharded path end to end (tests/test_cli_gpu.py); the default is one GPU per rank over RCCL
        if os.environ.get('PCC_DIST_SAME_GPU'):
            local_rank = 0
        torch.cuda.set_device(local_rank)
        if os.environ.get('PCC_DIST_BACKEND', 'nccl') == 'gloo':
            dist.init_process_group('gloo')
        else:
            dist.init_process_group('nccl', device_id=torch.device('cuda', local_rank))
    rank, world = sharding.world_info()
    assert not (args.debug and world > 1), '--debug checks every intermediate of every block: run it on one GPU'
    sess = ops.get_context(torch.device('cuda', local_rank))
    T.mark('context', sess.device)

    model = ModelConfigType[args.model_config].build(data_format=args.data_format, batch_size=args.batch_size, precision=args.precision)
    compressed_data = []
    for file in args.input_files:
        check_numerics_tag(read_gzip_tag(file), sess.numerics_tag(args.precision), ignore=args.ignore_numerics_tag)
        with gzip.open(file, 'rb') as f:
            compressed_data.append(load_compressed_file(f))
    T.mark('container_read_gunzip')
    model.decompress()
    model.restore(args.checkpoint_dir)
    T.mark('checkpoint_restore')
    if T.enabled():
        model._codec(sess)
        T.mark('weights_repack_upload', sess.device)

    for i, ((resolution, level, binstr, blocks), ori_file, output_file) in enumerate(
            zip(compressed_data, args.input_files, args.output_files)):
        logger.info(f'{i}/{len(args.input_files)} - Writing {ori_file} to {output_file} with {len(blocks)} blocks')
        x_shape = np.array([resolution, resolution, resolution], dtype=np.uint32) // (2 ** level)
        dec_blocks, debug_t_list = model.decompress_blocks(sess, blocks, x_shape, debug=args.debug)
        T.mark('decompress_blocks', sess.device)
        if args.debug and rank == 0:
            dec_blocks_enc = read_pcs(len(blocks), ori_file + '.enc.blocks')
            debug_data = np.load(ori_file + '.enc.data.npz', allow_pickle=True)
            data_list_enc, debug_t_list_enc = debug_data['data'], debug_data['debug_t_list']
            for j, (db, dbe) in enumerate(zip(dec_blocks, dec_blocks_enc)):
                assert [bytes(s) for s in blocks[j][0]] == [bytes(s) for s in data_list_enc[j][0]]
                for key in debug_t_list[j]:
                    np.testing.assert_array_equal(debug_t_list[j][key], debug_t_list_enc[j][key],
                                                  err_msg=f'block {j}: intermediate {key} differs between encoder and decoder')
                np.testing.assert_equal(db, dbe.astype(np.float32).reshape(-1, 3))
            logger.info(f'{i}/{len(args.input_files)} - all {len(blocks)} blocks verified against the encoder dumps')
        if rank != 0:
            continue
        bbox_max = x_shape * (2 ** level)
        dec_blocks = departition_octree(dec_blocks, binstr, [0, 0, 0], bbox_max, level)
        pa = np.vstack(dec_blocks) if len(dec_blocks) else np.zeros((0, 3))
        if os.path.split(output_file)[0]:
            os.makedirs(os.path.split(output_file)[0], exist_ok=True)
        pc_io.write_df(output_file, pc_io.pa_to_df(pa))
        T.mark('departition_ply_write')
    logger.info('Finished')
    T.dump()
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()


def build_parser():
    parser = argparse.ArgumentParser(prog='decompress_octree.py', description='Decompress a file.',
                                     formatter_class=argparse.ArgumentDefaultsHelpFormatter)
    parser.add_argument('--input_files', nargs='+', help='Input files.', required=True)
    parser.add_argument('--output_files', nargs='+', help='Output files.', required=True)
    parser.add_argument('--checkpoint_dir', help='Directory where to save/load model checkpoints.', required=True)
    parser.add_argument('--model_config', help='Model used: c1, c2, c3, c3p.', required=True)
    parser.add_argument('--num_filters', type=int, default=32, help='Number of filters per layer (ignored).')
    parser.add_argument('--data_format', default='channels_first', help='Data format used: channels_first or channels_last')
    parser.add_argument('--debug', default=False, action='store_true', help='Use debug data to check results.')
    parser.add_argument('--batch_size', type=int, default=32, help='Blocks resident on the GPU per pass (new).')
    parser.add_argument('--ignore_numerics_tag', default=False, action='store_true',
                        help='Decode although the stream was written under other codec numerics (kernel family / PCC_* switches / precision) (new).')
    parser.add_argument('--precision', default='fp32', choices=['fp32', 'fp16'],
                        help='fp16: fp16 matrix instructions with fp32 accumulation on the conv layers (new; must match between '
                             'compress and decompress).')
    return parser


if __name__ == '__main__':
    from . import want_hw_queues
    want_hw_queues()        # before torch (the HIP runtime) loads: decompress() imports it
    decompress(build_parser().parse_args())
    # everything is written and closed: leave without the interpreter / runtime teardown (0.4 - 0.5 s of a 2 - 3 s process,
    # profiles/r06_cli_wallclock.md); PCC_CLI_CLEAN_EXIT=1 keeps the ordinary exit
    if not os.environ.get('PCC_CLI_CLEAN_EXIT'):
        import sys
        logging.shutdown()
        sys.stdout.flush(); sys.stderr.flush()
        os._exit(0)
