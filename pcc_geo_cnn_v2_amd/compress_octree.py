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
import json
import logging
import os

import numpy as np

logging.basicConfig(level=logging.INFO,
                    format='%(asctime)s.%(msecs)03d %(levelname)s %(module)s - %(funcName)s: %(message)s',
                    datefmt='%Y-%m-%d %H:%M:%S')
logger = logging.getLogger(__name__)


def _dump_clouds(folder, clouds):
    """--debug: one PLY per block under `folder` (0.ply, 1.ply, ...)."""
    from .utils import pc_io
    os.makedirs(folder, exist_ok=True)
    for n, cloud in enumerate(clouds):
        pc_io.write_df(os.path.join(folder, f'{n}.ply'), pc_io.pa_to_df(cloud))


class _Cloud:
    """One input file of the command line and the files its rate points go to (one per optimisation metric)."""
    __slots__ = ('source', 'targets', 'decoded')

    def __init__(self, source, targets, decoded):
        self.source, self.targets, self.decoded = source, targets, decoded


def _plan(args):
    """Command-line contract of /root/reference/src/compress_octree.py:131-183 -> (clouds, with_normals).  Every input file owns
    len(opt_metrics) consecutive entries of --output_files (and of --dec_files when given)."""
    from .model_configs import ModelConfigType
    from .utils.pc_metric import validate_opt_metrics
    if args.resolution <= 0:
        raise AssertionError('resolution must be positive')
    if args.data_format not in ('channels_first', 'channels_last'):
        raise AssertionError(f'unknown data_format {args.data_format}')
    if args.model_config not in ModelConfigType.keys():
        raise AssertionError(f'unknown model_config {args.model_config}: one of {list(ModelConfigType.keys())}')
    with_normals = args.input_normals is not None
    # the reference's own default '--opt_metrics d1_psnr' is not in avail_opt_metrics, so its CLI asserts unless the flag is
    # given (compress_octree.py:37,154).  psnr is monotone in mse (pc_metric.py:55), so '<g>_psnr' is taken as '<g>_mse' here
    metrics = []
    for name in args.opt_metrics:
        if name in ('d1_psnr', 'd2_psnr'):
            logger.warning(f"--opt_metrics {name}: not an optimisation metric of the reference; using the equivalent {name[:3]}mse")
            name = name[:3] + 'mse'
        metrics.append(name)
    args.opt_metrics = metrics
    validate_opt_metrics(metrics, with_normals=with_normals)
    per_cloud = len(metrics) if len(metrics) > 1 else 1
    n_in = len(args.input_files)
    if len(args.output_files) != per_cloud * n_in:
        raise AssertionError(f'{n_in} input file(s) x {per_cloud} metric(s) need {per_cloud * n_in} output files, got {len(args.output_files)}')
    if per_cloud > 1 and len(args.input_normals or ()) * per_cloud != len(args.output_files):
        raise AssertionError('several optimisation metrics need one normals file per input file')
    if args.dec_files is not None and len(args.dec_files) != per_cloud * n_in:
        raise AssertionError(f'--dec_files: expected {per_cloud * n_in} paths, got {len(args.dec_files)}')
    clouds = []
    for k, source in enumerate(args.input_files):
        span = slice(k * per_cloud, (k + 1) * per_cloud)
        targets = list(args.output_files[span])
        if len(set(targets)) != len(targets):
            raise AssertionError(f'{targets} should have no duplicates')
        clouds.append(_Cloud(source, targets, None if args.dec_files is None else list(args.dec_files[span])))
    return clouds, with_normals


def _join_process_group():
    """-> (rank, world, local device index).  Under torch.distributed.run the blocks are sharded over the ranks."""
    import torch
    from . import sharding
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if int(os.environ.get('WORLD_SIZE', '1')) > 1:
        import torch.distributed as dist
        # PCC_DIST_BACKEND=gloo + PCC_DIST_SAME_GPU=1: every rank on GPU 0 with host-side collectives -- lets a 1-GPU box run the
        # sharded path end to end (tests/test_cli_gpu.py); the default is one GPU per rank over RCCL
        if os.environ.get('PCC_DIST_SAME_GPU'):
            local = 0
        torch.cuda.set_device(local)
        if os.environ.get('PCC_DIST_BACKEND', 'nccl') == 'gloo':
            dist.init_process_group('gloo')
        else:
            dist.init_process_group('nccl', device_id=torch.device('cuda', local))
    rank, world = sharding.world_info()
    return rank, world, local


def _block_grid(resolution, level, data_format):
    """Octree geometry: the bounding box of the whole cloud and the dense shape of one leaf block."""
    from .utils import pc_io
    _, _, shape = pc_io.get_shape_data(resolution, data_format)
    spatial = slice(1, 4) if data_format == 'channels_first' else slice(0, 3)
    box = shape[spatial].copy()
    shape[spatial] = shape[spatial] // (1 << level)
    return box, shape


def _write_rate_point(target, decoded_path, binstr, streams, info, args, blocks, debug_t_list):
    """The files of one rate point: <target> (gzip'd container), <target>.enc.metric.json, optionally the decoded cloud and the
    --debug dumps (/root/reference/src/compress_octree.py:109-125 lists them)."""
    from .model_syntax import save_compressed_file, write_tagged_gzip
    from .utils import pc_io
    folder = os.path.dirname(target)
    if folder:
        os.makedirs(folder, exist_ok=True)
    payload = save_compressed_file(binstr, streams, args.resolution, args.octree_level, strict=True)
    write_tagged_gzip(target, payload, info['numerics_tag'])      # = gzip.open(target, 'wb').write(payload) + the tag in the member header
    with open(target + '.enc.metric.json', 'w') as fh:
        # the reference's keys (floats) + the tag of the kernels that computed sigma-hat (a string: a tool that re-gzips the payload drops the
        # header comment, this file keeps it; readers of the reference's JSON iterate over known metric names)
        json.dump(dict({name: float(val) for name, val in info['metrics'].items()}, codec_numerics=info['numerics_tag']), fh, sort_keys=True, indent=4)
    if decoded_path is not None:
        pc_io.write_df(decoded_path, pc_io.pa_to_df(info['blocks_full']))
    if args.debug:
        pc_io.write_df(target + '.enc.ply', pc_io.pa_to_df(info['blocks_full']))
        _dump_clouds(target + '.ori.blocks', blocks)
        _dump_clouds(target + '.enc.blocks', info['x_hat_list'])
        _dump_clouds(target + '.enc.blocks.depart', info['blocks_depart'])
        np.savez_compressed(target + '.enc.data.npz', data=np.array(streams, dtype=object), debug_t_list=np.array(debug_t_list, dtype=object))


def compress(args):
    from .utils import cli_timing as T
    T.mark('process_start_to_main')
    import torch
    from . import ops
    from .model_configs import ModelConfigType
    from .utils import pc_io
    from .utils.octree_coding import partition_octree
    T.mark('imports')

    clouds, with_normals = _plan(args)
    rank, world, local = _join_process_group()
    if args.debug and world > 1:
        raise AssertionError('--debug dumps every intermediate of every block: run it on one GPU')
    sess = ops.get_context(torch.device('cuda', local))        # what tf.Session is to the reference (compress_octree.py:84)
    T.mark('context', sess.device)

    geometry = pc_io.load_points(args.input_files, batch_size=args.read_batch_size)
    if with_normals:
        geometry = [np.hstack((xyz, pc_io.load_normals(path))) for xyz, path in zip(geometry, args.input_normals)]
    T.mark('ply_read')
    box, block_shape = _block_grid(args.resolution, args.octree_level, args.data_format)
    logger.info('Performing octree partitioning')
    partitions = [partition_octree(cloud, [0, 0, 0], box, args.octree_level) for cloud in geometry]
    T.mark('partition')
    logger.info(f'Processing resolution {args.resolution} with octree level {args.octree_level} resulting in dense_tensor_shape '
                f'{block_shape} and {sum(len(blocks) for blocks, _ in partitions)} blocks')

    model = ModelConfigType[args.model_config].build(data_format=args.data_format, batch_size=args.batch_size, precision=args.precision)
    model.d2_search = getattr(args, 'd2_search', None)      # per model, not a module global
    model.compress(np.concatenate(((1,), block_shape)))
    model.restore(args.checkpoint_dir)      # asserts 'Checkpoint ... was not found' like compress_octree.py:91
    T.mark('checkpoint_restore')
    if T.enabled():      # the weight repack + upload happens on the first codec call: make it a phase of its own
        model._codec(sess)
        T.mark('weights_repack_upload', sess.device)

    want_points = clouds[0].decoded is not None or args.debug
    for cloud, points, (blocks, binstr) in zip(clouds, geometry, partitions):
        logger.info(f'Starting {cloud.source} to {", ".join(cloud.targets)} with {len(blocks)} blocks')
        streams, infos, debug_t_list = model.compress_blocks(
            sess, blocks, binstr, points, args.resolution, args.octree_level, with_normals=with_normals, opt_metrics=args.opt_metrics,
            max_deltas=args.max_deltas, fixed_threshold=args.fixed_threshold, debug=args.debug, need_points=want_points)
        T.mark('compress_blocks', sess.device)
        if rank == 0:       # the other ranks only took part in the collectives
            if len(streams) != len(cloud.targets):
                raise AssertionError(f'{len(streams)} rate points for {len(cloud.targets)} output files')
            for n, target in enumerate(cloud.targets):
                infos[n]['numerics_tag'] = sess.numerics_tag(args.precision)
                _write_rate_point(target, None if cloud.decoded is None else cloud.decoded[n], binstr, streams[n], infos[n], args, blocks, debug_t_list)
            logger.info(f'Finished {cloud.source} to {", ".join(cloud.targets)} with {len(blocks)} blocks')
        T.mark('container_gzip_write')
    T.dump()
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
    parser.add_argument('--d2_search', default=None, choices=['gpu', 'kdtree'],
                        help='Where the d2_* statistics of the adaptive threshold search come from (new): gpu = nearest-index transforms, ties '
                             'between equidistant neighbours go to the lowest (x, y, z); kdtree = scipy KD-trees on the host, the '
                             "reference's own picks and decisions (8-18x slower per cloud).  Default: kdtree (PCC_D2_GPU=1 = gpu).")
    parser.add_argument('--precision', default='fp32', choices=['fp32', 'fp16'],
                        help='fp16: fp16 matrix instructions with fp32 accumulation on the conv layers (new; must match between '
                             'compress and decompress).')
    return parser


if __name__ == '__main__':
    from . import want_hw_queues
    want_hw_queues()        # before torch (the HIP runtime) loads: compress() imports it
    compress(build_parser().parse_args())
    # everything is written and closed: leave without the interpreter / runtime teardown (0.4 - 0.5 s of a 2 - 3 s process,
    # profiles/r06_cli_wallclock.md); PCC_CLI_CLEAN_EXIT=1 keeps the ordinary exit
    if not os.environ.get('PCC_CLI_CLEAN_EXIT'):
        import sys
        logging.shutdown()
        sys.stdout.flush(); sys.stderr.flush()
        os._exit(0)
