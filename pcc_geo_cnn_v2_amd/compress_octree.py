"""Encoder CLI -- same flags and files as /root/reference/src/compress_octree.py:130-185.

  python -m pcc_geo_cnn_v2_amd.compress_octree --input_files a.ply --output_files a.ply.bin \\
         --checkpoint_dir models/c3p/1.00e-04 --model_config c3p --resolution 1024 --octree_level 4 \\
         [--dec_files a.dec.ply] [--fixed_threshold] [--opt_metrics d1_mse] [--max_deltas inf] [--debug]

Differences: `--checkpoint_dir` holds `model.npz` (this framework's weight container) instead of a TF1
checkpoint; `--batch_size` (blocks resident per GPU pass) is new.  Under
`python -m torch.distributed.run --nproc-per-node N` the blocks are sharded over N GPUs and rank 0 writes
the files.  `--num_filters` is accepted and ignored, like in the reference.
"""
import argparse
import gzip
import json
import logging
import os

import numpy as np

logging.basicConfig(level=logging.INFO,
                    format='%(asctime)s.%(msecs)03d %(levelname)s %(module)s - %(funcName)s: %(message)s',
                    datefmt='%Y-%m-%d %H:%M:%S')
logger = logging.getLogger(__name__)


def write_pcs(pcs, folder):
    from .utils import pc_io
    os.makedirs(folder, exist_ok=True)
    for j, points in enumerate(pcs):
        pc_io.write_df(os.path.join(folder, f'{j}.ply'), pc_io.pa_to_df(points))


def compress(args):
    import torch
    from . import ops, sharding
    from .model_configs import ModelConfigType
    from .model_syntax import save_compressed_file
    from .utils import pc_io
    from .utils.octree_coding import partition_octree
    from .utils.pc_metric import validate_opt_metrics

    assert args.resolution > 0, 'resolution must be positive'
    assert args.data_format in ['channels_first', 'channels_last']
    with_normals = args.input_normals is not None
    # the reference's own default '--opt_metrics d1_psnr' is not in avail_opt_metrics, so its CLI asserts unless the flag is
    # given (compress_octree.py:37,154).  psnr is monotone in mse (pc_metric.py:55), so '<g>_psnr' is taken as '<g>_mse' here
    for k, m in enumerate(args.opt_metrics):
        if m in ('d1_psnr', 'd2_psnr'):
            args.opt_metrics[k] = m[:3] + 'mse'
            logger.warning(f"--opt_metrics {m}: not an optimisation metric of the reference; using the equivalent {args.opt_metrics[k]}")
    validate_opt_metrics(args.opt_metrics, with_normals=with_normals)
    files_mult = 1
    if len(args.opt_metrics) > 1:
        files_mult *= len(args.opt_metrics)
        assert files_mult * len(args.input_files) == len(args.output_files)
        assert files_mult * len(args.input_normals) == len(args.output_files)
    else:
        assert files_mult * len(args.input_files) == len(args.output_files)
    decode_files = args.dec_files is not None
    if decode_files:
        assert files_mult * len(args.input_files) == len(args.dec_files)
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
    assert not (args.debug and world > 1), '--debug dumps every intermediate of every block: run it on one GPU'
    sess = ops.get_context(torch.device('cuda', local_rank))  # replaces tf.Session (compress_octree.py:84)

    p_min, p_max, dense_tensor_shape = pc_io.get_shape_data(args.resolution, args.data_format)
    points = pc_io.load_points(args.input_files, batch_size=args.read_batch_size)
    if with_normals:
        normals = [pc_io.load_normals(x) for x in args.input_normals]
        points = [np.hstack((p, n)) for p, n in zip(points, normals)]

    logger.info('Performing octree partitioning')
    bbox_min = [0, 0, 0]
    if args.data_format == 'channels_first':
        bbox_max = dense_tensor_shape[1:].copy()
        dense_tensor_shape[1:] = dense_tensor_shape[1:] // (2 ** args.octree_level)
    else:
        bbox_max = dense_tensor_shape[:3].copy()
        dense_tensor_shape[:3] = dense_tensor_shape[:3] // (2 ** args.octree_level)
    blocks_list, binstr_list = zip(*[partition_octree(p, bbox_min, bbox_max, args.octree_level) for p in points])
    n_total = sum(len(b) for b in blocks_list)
    logger.info(f'Processing resolution {args.resolution} with octree level {args.octree_level} resulting in '
                f'dense_tensor_shape {dense_tensor_shape} and {n_total} blocks')

    x_shape = np.concatenate(((1,), dense_tensor_shape))
    model = ModelConfigType[args.model_config].build(data_format=args.data_format, batch_size=args.batch_size, precision=args.precision)
    model.compress(x_shape)
    model.restore(args.checkpoint_dir)  # asserts 'Checkpoint ... was not found' like compress_octree.py:91

    for i in range(len(args.input_files)):
        ori_file, cur_points, blocks, binstr = [x[i] for x in (args.input_files, points, blocks_list, binstr_list)]
        cur_output_files = [args.output_files[i * files_mult + j] for j in range(files_mult)]
        if decode_files:
            cur_dec_files = [args.dec_files[i * files_mult + j] for j in range(files_mult)]
        assert len(set(cur_output_files)) == len(cur_output_files), f'{cur_output_files} should have no duplicates'
        logger.info(f'Starting {ori_file} to {", ".join(cur_output_files)} with {len(blocks)} blocks')
        data_list, data, debug_t_list = model.compress_blocks(sess, blocks, binstr, cur_points, args.resolution,
                                                              args.octree_level, with_normals=with_normals,
                                                              opt_metrics=args.opt_metrics, max_deltas=args.max_deltas,
                                                              fixed_threshold=args.fixed_threshold, debug=args.debug,
                                                              need_points=decode_files or args.debug)
        if rank != 0:
            continue
        assert len(data_list) == files_mult
        for j in range(len(cur_output_files)):
            of, cur_data_list, cur_data = [x[j] for x in (cur_output_files, data_list, data)]
            if os.path.split(of)[0]:
                os.makedirs(os.path.split(of)[0], exist_ok=True)
            with gzip.open(of, 'wb') as f:
                f.write(save_compressed_file(binstr, cur_data_list, args.resolution, args.octree_level, strict=True))
            if decode_files:
                pc_io.write_df(cur_dec_files[j], pc_io.pa_to_df(cur_data['blocks_full']))
            with open(of + '.enc.metric.json', 'w') as f:
                json.dump({k: float(v) for k, v in cur_data['metrics'].items()}, f, sort_keys=True, indent=4)
            if args.debug:
                pc_io.write_df(of + '.enc.ply', pc_io.pa_to_df(cur_data['blocks_full']))
                write_pcs(blocks, of + '.ori.blocks')
                write_pcs(cur_data['x_hat_list'], of + '.enc.blocks')
                write_pcs(cur_data['blocks_depart'], of + '.enc.blocks.depart')
                np.savez_compressed(of + '.enc.data.npz', data=np.array(cur_data_list, dtype=object),
                                    debug_t_list=np.array(debug_t_list, dtype=object))
        logger.info(f'Finished {ori_file} to {", ".join(cur_output_files)} with {len(blocks)} blocks')
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()


def build_parser():
    from .utils.pc_metric import avail_opt_metrics
    parser = argparse.ArgumentParser(prog='compress_octree.py', description='Compress a file.',
                                     formatter_class=argparse.ArgumentDefaultsHelpFormatter)
    parser.add_argument('--input_files', nargs='+', help='Input files.', required=True)
    parser.add_argument('--output_files', nargs='+', required=True,
                        help='Output files. If input normals are provided, specify two output files per input file.')
    parser.add_argument('--input_normals', nargs='+',
                        help='Input normals. If provided, two output paths are needed for each input file for D1 and D2 optimization.')
    parser.add_argument('--dec_files', nargs='*',
                        help='Decoded files. Allows compression/decompression in a single execution.')
    parser.add_argument('--checkpoint_dir', help='Directory where to save/load model checkpoints.', required=True)
    parser.add_argument('--model_config', help='Model used: c1, c2, c3, c3p.', required=True)
    # the reference's default (compress_octree.py:156); translated to d1_mse in compress(), see there
    parser.add_argument('--opt_metrics', nargs='+', default=['d1_psnr'],
                        help=f'Optimization metrics used. Available: {avail_opt_metrics}')
    parser.add_argument('--max_deltas', nargs='+', default=[np.inf], type=float, help='Max deltas tested during optimization.')
    parser.add_argument('--fixed_threshold', default=False, action='store_true', help='Enable fixed thresholding.')
    parser.add_argument('--read_batch_size', type=int, default=1, help='Batch size for parallel reading.')
    parser.add_argument('--resolution', type=int, help='Dataset resolution.', default=64)
    parser.add_argument('--octree_level', type=int, help='Octree level.', default=4)
    parser.add_argument('--num_filters', type=int, default=32, help='Number of filters per layer (ignored, as in the reference).')
    parser.add_argument('--data_format', default='channels_first', help='Data format used: channels_first or channels_last')
    parser.add_argument('--debug', default=False, action='store_true', help='Output debug data for point cloud.')
    parser.add_argument('--batch_size', type=int, default=32, help='Blocks resident on the GPU per pass (new).')
    parser.add_argument('--precision', default='fp32', choices=['fp32', 'fp16'],
                        help='fp16: fp16 matrix instructions with fp32 accumulation on the conv layers (new; must match between '
                             'compress and decompress).')
    return parser


if __name__ == '__main__':
    compress(build_parser().parse_args())
