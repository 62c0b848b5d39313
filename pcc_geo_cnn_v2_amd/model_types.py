"""Compression models -- the interface of /root/reference/src/model_types.py:179-416
(CompressionModel / CompressionModelV1 / CompressionModelV2: compress(), decompress(),
compress_blocks(), decompress_blocks()) on top of the HIP kernels.

MI355X-first differences (results identical):
  * blocks are processed `batch_size` at a time, resident in HBM, instead of one sess.run per block
    (reference: model_types.py:192-198,224-230);
  * sparse_to_dense, thresholding and np.argwhere run on the GPU (voxelize / threshold_compact);
  * the range coder runs on host threads, one stream per block, overlapped with the synthesis
    transform of the same (encoder) or previous (decoder) chunk;
  * the encoder obtains z_hat / y_hat from the quantiser directly instead of range-decoding the
    string it just produced (model_types.py:383,387): the values are identical by construction.
`sess` in the reference's signatures is an `ops.Context` here (None = the default GPU context).
"""
import logging
import os
import pprint
import sys
import time
from enum import Enum

import numpy as np
import torch
from scipy.spatial import cKDTree

from . import _lib as L
from . import model_transforms as MT
from . import ops
from .entropy_models import EntropyBottleneck, GaussianConditional, scale_table
from .model_opt import d1_tallies_gpu, d12_tallies_gpu, d2_on_gpu, decide_from_tallies, gpu_search_supported, metric_names
from .model_transforms import TransformType
from .utils.octree_coding import departition_octree
from .utils.pc_metric import cloud_metrics_batch, finish_metrics

logger = logging.getLogger(__name__)

CHECKPOINT_FILE = 'model.npz'


def sparse_to_dense(block, x_shape, data_format):
    """Host version of the dense occupancy scatter (model_types.py:108-114), kept for API parity and as the test oracle of
    ops.voxelize, which the codec uses.  x_shape is the 5-D shape of ONE block in `data_format`."""
    assert data_format in ('channels_first', 'channels_last')
    grid = np.zeros(x_shape, dtype=np.float32)
    occupancy = grid[0, 0] if data_format == 'channels_first' else grid[0, ..., 0]      # (D, H, W) view of the single channel
    occupancy[tuple(np.asarray(block)[:, :3].astype(np.uint32).T)] = 1.0                 # out-of-range coordinate: IndexError
    return grid


def get_normals_if(x, with_normals):
    return x[:, x.shape[1] - 3:x.shape[1]] if with_normals else None


def rank_candidates(names, cand_metrics, opt_groups=('d1', 'd2')):
    """The selection rule of model_types.py:128-176 on already-computed metrics: per optimisation group ('d1', 'd2') the
    candidate -- among those whose metric name starts with the group -- with the highest '{group}_psnr'; an empty decoded
    cloud (metrics None) scores -inf.  Returns [(group, winning candidate index, its metrics)].  Shared by the single-process
    and the sharded encoder (the latter feeds all_reduce'd metrics, so every rank takes the same decision)."""
    picks = []
    for group in opt_groups:
        members = [i for i, n in enumerate(names) if n.startswith(group)]
        if not members:
            continue
        key = f'{group}_psnr'
        score = {i: (-np.inf if cand_metrics[i] is None else cand_metrics[i][key]) for i in members}
        winner = members[int(np.argmax([score[i] for i in members]))]
        picks.append((group, winner, cand_metrics[winner] if cand_metrics[winner] is not None else {key: -np.inf}))
        logger.info(f'Group {group} : {key} best idx {winner} {names[winner]}\n' +
                    pprint.pformat({names[i]: f'{score[i]:.2f}' for i in members}))
    return picks


def select_best_per_opt_metric(binstr, x_hat_list, level, opt_metrics, points, resolution, with_normals,
                               opt_groups=('d1', 'd2'), tree=None):
    """Per optimisation group, which candidate reconstruction of the whole cloud to keep (the reference's function of the same
    name, model_types.py:128-176; results pinned by tests/golden/select_best.npz).  x_hat_list[m] = the decoded blocks of
    candidate m (block-local coordinates).  Returns one dict per non-empty group: 'idx', 'metrics', 'x_hat_list',
    'blocks_depart' (blocks in cloud coordinates), 'blocks_full' (one array)."""
    assert len(opt_metrics) == len(x_hat_list), f'lengths of opt_metrics {len(opt_metrics)} and x_hat_list' + \
                                                f' {len(x_hat_list)} should be equal'
    grouped = [m for m, n in enumerate(opt_metrics) if any(n.startswith(g) for g in opt_groups)]
    placed = {m: departition_octree(x_hat_list[m], binstr, [0, 0, 0], [resolution] * 3, level) for m in grouped}
    clouds = {m: np.vstack(placed[m]) for m in grouped}
    original = points[:, :3]
    # `tree`: the KD-tree over the original points when the caller built it meanwhile (compress_blocks: beside the block loop)
    scored = cloud_metrics_batch(original, [clouds[m] for m in grouped], resolution - 1, get_normals_if(points, with_normals),
                                 tree if tree is not None else cKDTree(original))
    cand_metrics = [None] * len(opt_metrics)
    for m, met in zip(grouped, scored):
        cand_metrics[m] = met
    return [{'idx': m, 'metrics': met, 'x_hat_list': x_hat_list[m], 'blocks_depart': placed[m], 'blocks_full': clouds[m]}
            for _, m, met in rank_candidates(opt_metrics, cand_metrics, opt_groups)]


_SIDE_STREAMS = {}


def _usable_cores():
    """Cores this process may really use: the affinity mask, capped by the cgroup CPU quota (a container that sees 256 cores may
    be throttled to 16 CPUs' worth of time per period -- running more threads than that gets the whole process paused)."""
    return ops.usable_cores()


class _Immediate:
    """A future-like wrapper that runs its function when the result is asked for (the caller's thread)."""

    def __init__(self, fn):
        self.fn = fn

    def result(self):
        return self.fn()


def _host_dtypes(levels=64):
    """(symbol dtype, CDF-row dtype) of the host staging buffers: int16 / uint8 (uint8 rows cover scale tables of up to 256
    levels; larger tables keep int32 rows); PCC_WIDE_SYMBOLS=1 keeps int32 for both (A/B runs)."""
    if os.environ.get('PCC_WIDE_SYMBOLS'):
        return torch.int32, torch.int32
    return torch.int16, (torch.uint8 if levels <= 256 else torch.int32)


class _Pinned:
    """Cache of pinned host staging buffers keyed by (tag, shape, dtype)."""

    def __init__(self):
        self._b = {}
        self._rings = {}

    def get(self, tag, shape, dtype):
        key = (tag, tuple(shape), dtype)
        if key not in self._b:
            self._b[key] = torch.empty(tuple(shape), dtype=dtype, pin_memory=True)
        return self._b[key]

    def ring(self, tag, shape, dtype, depth=4):
        """Next buffer of a ring of `depth` pinned buffers (the decoder's staging buffers: a chunk's buffer is still the source
        of an asynchronous host->device copy while the host already fills the next chunk's).  Returns (buffer, release):
        call release(stream) after enqueuing the last device operation that reads the buffer; the ring waits for that event
        before it hands the buffer out again -- no pinned allocation (a device-synchronising call) in the steady state."""
        key = (tag, tuple(shape), dtype)
        r = self._rings.setdefault(key, {'next': 0, 'buf': [None] * depth, 'busy': [None] * depth})
        i = r['next']
        r['next'] = (i + 1) % depth
        if r['busy'][i] is not None:
            r['busy'][i].synchronize()
            r['busy'][i] = None
        if r['buf'][i] is None:
            r['buf'][i] = torch.empty(tuple(shape), dtype=dtype, pin_memory=True)

        def release(stream):
            ev = torch.cuda.Event()
            ev.record(stream)
            r['busy'][i] = ev
        return r['buf'][i], release


class CompressionModel:
    def __init__(self, n_thresholds=2 ** 8, data_format='channels_first', batch_size=32,
                 round_mode=L.PCC_ROUND_FLOOR_HALF, coder_threads=0, seed=42, precision='fp32', search_threads=0):
        assert precision in ('fp32', 'fp16'), "precision: 'fp32' (the reference's arithmetic) or 'fp16' (fp16 matrix instructions, fp32 accumulate)"
        self.precision = precision
        self.thresholds = np.linspace(0, 1.0, n_thresholds)
        self.data_format = data_format
        self.batch_size = int(batch_size)
        self.round_mode = round_mode
        self.coder_threads = coder_threads
        self.search_threads = search_threads      # worker processes of the host KD-tree threshold search (0 = min(cores, 64))
        self.seed = seed
        self.x_shape = None
        self._pinned = _Pinned()
        self._dev_cache = {}
        self._host_cache = {}

    # ------------------------------------------------------------------ helpers
    def _bind_transforms(self, **kinds):
        """`<kind>_transform_class` from the TransformType members of the ctor; `<kind>_transform` (the instance) is created by
        compress() / decompress()."""
        for kind, member in kinds.items():
            setattr(self, f'{kind}_transform_class', member.value)
            setattr(self, f'{kind}_transform', None)

    def _ctx(self, sess):
        ctx = sess if isinstance(sess, (ops.Context, ops._ContextView)) else ops.get_context(None)
        # BASELINE.json configs[4]: the fp16 mode.  Encoder and decoder must agree on it (the decoder recomputes sigma_hat):
        # like the checkpoint, it is part of the codec configuration, not of the stream.  A VIEW of the context carries the
        # flag: the caller's context is not modified.
        return ctx.view(L.PCC_CONV_F16 if self.precision == 'fp16' else 0)

    def _dev(self, ctx, name, arr):
        key = (ctx.device.index, name)
        if key not in self._dev_cache:
            self._dev_cache[key] = torch.from_numpy(np.ascontiguousarray(arr)).to(ctx.device)
        return self._dev_cache[key]

    def _spatial(self, x_shape):
        x_shape = [int(v) for v in x_shape]
        if len(x_shape) == 3:
            return tuple(x_shape)
        assert len(x_shape) == 5
        return tuple(x_shape[2:5]) if self.data_format == 'channels_first' else tuple(x_shape[1:4])

    def _side_stream(self, ctx, which='_copy_stream'):
        """Copy streams beside the main one.  Work on one stream runs in order, so copies with different dependencies get
        different streams: '_copy_stream' (encoder symbols and decoded points to the host: each waits for an event of the main
        stream), '_up_stream' (decoder symbols to the device: no GPU-side dependency, they run as soon as the host has decoded
        them), '_idx_stream' (the decoder's CDF-row indexes to the host)."""
        # one set per device for the whole process (streams are a runtime resource: every model on the device shares them)
        key = (ctx.device.index, which)
        if key not in _SIDE_STREAMS:
            _SIDE_STREAMS[key] = torch.cuda.Stream(ctx.device)
        return _SIDE_STREAMS[key]

    def _staging(self, ctx, slot, B, y_dhw, z_dhw=None):
        """Per pipeline slot: the device + pinned staging buffers of one encode (ops.SymbolStaging), cached."""
        sym_t, row_t = _host_dtypes(len(getattr(self, 'scale_table', ())) or 64)
        F = self.num_filters
        key = (ctx.device.index, slot, B, tuple(y_dhw), None if z_dhw is None else tuple(z_dhw), sym_t, self.data_format)
        cache = self.__dict__.setdefault('_stagings', {})
        if key not in cache:
            cache[key] = ops.SymbolStaging(ctx.device, B, self._stream_shape(B, y_dhw, F),
                                           None if z_dhw is None else self._stream_shape(B, z_dhw, F), F, sym_t, row_t,
                                           self.data_format == 'channels_first')
        return cache[key]

    def _ship(self, ctx, staging, ready=None):
        """ONE device->pinned-host copy of a packed staging buffer on the side stream (no kernel runs there: the permutation
        into stream order, the narrowing and the max|symbol| tiles were written in order on the main stream by the library);
        returns the event the host has to wait for.  `ready`: event after which the staging buffer is final (default: now,
        on the main stream)."""
        main = torch.cuda.current_stream(ctx.device)
        side = self._side_stream(ctx)
        if ready is None:
            ready = torch.cuda.Event()
            ready.record(main)
        with torch.cuda.stream(side):
            side.wait_event(ready)
            staging.copy_out()
            done = torch.cuda.Event()
            done.record(side)
        return done

    # ---- symbol order of the range-coded streams.  tfc 1.3 codes each batch item's tensor flattened in ITS memory order:
    # with the reference's default data_format='channels_first' (model_types.py:180,254,377) that is (C,D,H,W), i.e.
    # channel-major streams; 'channels_last' gives (D,H,W,C).  Internally everything is NDHWC, so for channels_first the
    # int32 symbols / indexes are permuted on the GPU before they leave (and after they come back).
    def _to_stream_order(self, t):
        """(B,D,H,W,C) device tensor -> contiguous tensor in the stream's flattening order."""
        return t.permute(0, 4, 1, 2, 3).contiguous() if self.data_format == 'channels_first' else t

    def _from_stream_order(self, t):
        """inverse of _to_stream_order: returns (B,D,H,W,C) contiguous."""
        return t.permute(0, 2, 3, 4, 1).contiguous() if self.data_format == 'channels_first' else t

    def _stream_shape(self, B, dhw, C):
        return (B, C) + tuple(dhw) if self.data_format == 'channels_first' else (B,) + tuple(dhw) + (C,)

    def _eb_rows(self, n_per_block, C):
        """EntropyBottleneck CDF row (= channel) of every symbol of one stream: (index_list, index_mod) for the range coder."""
        if self.data_format != 'channels_first':
            return None, C                          # row = i % C
        key = ('eb_rows', n_per_block, C)
        if key not in self._host_cache:
            self._host_cache[key] = np.repeat(np.arange(C, dtype=np.int32), n_per_block // C)
        return self._host_cache[key], 0

    def _thr32(self, idx):
        # the reference compares float32 x_hat with a float64 scalar under numpy 1.18 value-based
        # casting, i.e. in float32
        return np.float32(self.thresholds[idx])

    def _voxelize(self, ctx, blocks, dhw):
        D, H, W = dhw
        B = len(blocks)
        pts = np.ascontiguousarray(np.concatenate([np.asarray(b)[:, :3] for b in blocks]).astype(np.uint32).astype(np.int32))
        # the reference's dense scatter raises IndexError on an out-of-range coordinate (model_types.py:108-114)
        assert pts.size == 0 or (pts.min() >= 0 and np.all(pts.max(0) < np.array(dhw))), \
            f'block-local coordinates outside the {dhw} grid'
        bof = np.concatenate([np.full(len(b), i, np.int32) for i, b in enumerate(blocks)])
        return ops.voxelize(ctx, torch.from_numpy(pts).to(ctx.device), torch.from_numpy(bof).to(ctx.device), B, D, H, W)

    def _search_pool(self, n_jobs):
        from .model_opt import HostSearchPool
        # default: the cores this process may run on, at most 64 (measured on a 256-thread box whose container gets far fewer: 64
        # workers 17.5 s per 190-block cloud with d2 metrics, 128 workers 23.6 s -- the KD-tree work is host-bound)
        usable = _usable_cores() * 4          # (KD-tree queries wait on memory: 64 workers on a 16-CPU quota measured best)
        want = max(1, min(n_jobs, self.search_threads or min(usable, 64)))
        pool = getattr(self, '_host_search_pool', None)
        if pool is None or len(pool.procs) < want:
            if pool is not None:
                pool.close()
            self._host_search_pool = pool = HostSearchPool(want)
        return pool

    def _helper_thread(self, name):
        """Single helper threads of roundtrip_stream besides the encoder's: 'ydec' (the decoder's y range-decode), 'gather' (point lists
        to the host).  One thread each: the jobs of a kind complete in submission order."""
        pools = self.__dict__.setdefault('_helper_pools', {})
        if name not in pools:
            from concurrent.futures import ThreadPoolExecutor
            pools[name] = ThreadPoolExecutor(max_workers=1, thread_name_prefix=f'pcc-{name}')
        return pools[name]

    def _coder_thread(self):
        """One helper thread for host range-coder work that may overlap the calling thread's (roundtrip_stream)."""
        if getattr(self, '_coder_pool', None) is None:
            from concurrent.futures import ThreadPoolExecutor
            self._coder_pool = ThreadPoolExecutor(max_workers=1, thread_name_prefix='pcc-coder')
        return self._coder_pool

    def _thr_tensor(self, ctx, thr_idx):
        """float32 thresholds of the blocks of a chunk as a device tensor."""
        # cached on the device: a pageable host->device copy here would block the host until the GPU drains
        # and serialise the pipeline
        thr_idx = [int(t) for t in thr_idx]
        if len(set(thr_idx)) == 1:     # fixed threshold: one cached vector per (value, batch)
            thr = self._dev(ctx, ('thr', thr_idx[0], len(thr_idx)), np.full(len(thr_idx), self._thr32(thr_idx[0]), np.float32))
        else:                          # adaptive thresholds vary per chunk: gather from the cached 256-entry table
            table = self._dev(ctx, 'thr_table', np.array([self._thr32(t) for t in range(len(self.thresholds))], np.float32))
            thr = table[torch.tensor(thr_idx, dtype=torch.int64).to(ctx.device, non_blocking=True)]
        return thr

    def _extract_points(self, ctx, x_hat, thr_idx, clip):
        """x_hat (B,D,H,W) device; thr_idx list of ints -> (xyz (B,cap,3), counts (B,)) device tensors."""
        return ops.threshold_compact(ctx, x_hat, self._thr_tensor(ctx, thr_idx), clip=clip)

    def _codec(self, ctx):
        """pcc_codec_desc of this model on ctx's GPU (the batched-graph ABI), or None when a transform is not one of the
        plain reference stacks (then the per-layer path is used)."""
        nets = {}
        for prefix, tr, _ in self._transforms():
            nets[prefix] = tr.network() if hasattr(tr, 'network') else None
            if nets[prefix] is None:
                return None
        key = (ctx.device.index,) + tuple(id(n) for n in nets.values())
        if getattr(self, '_codec_cache', (None,))[0] != key:
            v2 = isinstance(self, CompressionModelV2)
            med = self._dev(ctx, 'medians', self.entropy_bottleneck.medians)
            tab = self._dev(ctx, 'scale_table', self.conditional_bottleneck.scale_table_f32) if v2 else None
            # the NetworkWeights objects are kept alive with the entry: their id()s are the key and must not be recycled
            self._codec_cache = (key, ops.codec_desc(ctx, 2 if v2 else 1, self.num_filters, nets, med, tab, self.round_mode),
                                 list(nets.values()))
        return self._codec_cache[1][0]

    def _range_decode(self, table, strings, n, index_list, index_mod, sym_h):
        """Range-decodes one stream per block into the narrow pinned buffer `sym_h` (B, ...) in stream order; returns sym_h, or
        -- when a symbol does not fit it (OverflowError from the coder: never seen in practice) -- an int32 array of the same
        shape."""
        B = len(strings)
        try:
            # 2-D fast paths of the wrapper: sym_h (B, ...) is filled in place, the rows come as one 2-D tensor or one shared vector
            ops.range_decode_batch(table, strings, [n] * B, index_list, index_mod, self.coder_threads, out=sym_h.view(B, -1))
            return sym_h
        except OverflowError:
            if index_list is not None and not isinstance(index_list, (list, tuple)):     # the list form of the wrapper
                index_list = [index_list[b] for b in range(B)] if getattr(index_list, 'ndim', 1) >= 2 else [index_list] * B
            wide = ops.range_decode_batch(table, strings, [n] * B, index_list, index_mod, self.coder_threads)
            return torch.from_numpy(np.stack(wide).reshape(sym_h.shape))

    def _symbols_to_device(self, ctx, sym_host, release):
        """Stream-order host symbols -> the device, as they are (one host->device copy of the narrow integers: half the PCIe
        bytes of int32).  The copy runs on the SIDE stream -- a copy on the main stream would hold back every kernel queued
        behind it for its 20-60 us, and the decoder calls of a chunk are enqueued long before the GPU gets to them -- on a
        stream of their own (nothing there ever waits for the GPU), and the main stream only waits for the copy's event.  The library unpacks the symbols into the int32 (B,D,H,W,C) tensor inside the
        decoder call (pcc_symbol_io); the per-layer path calls ops.symbols_unpack."""
        main, side = torch.cuda.current_stream(ctx.device), self._side_stream(ctx, '_up_stream')
        with torch.cuda.stream(side):
            dev = sym_host.to(ctx.device, non_blocking=True)
            release(side)
            arrived = torch.cuda.Event()
            arrived.record(side)
        main.wait_event(arrived)
        dev.record_stream(main)
        return dev

    def _unpack(self, ctx, packed, dhw):
        return ops.symbols_unpack(ctx, packed, (packed.shape[0],) + tuple(dhw) + (self.num_filters,), self.data_format == 'channels_first')

    def _gather_points(self, xyz, counts, ctx=None, ready=None):
        """Point lists to the host.  When `ready` (an event recorded after the compaction kernels) is given, the
        copies run on the side stream and wait only for that event, so the host never drains the main queue."""
        if ready is None or ctx is None:
            cnt = counts.cpu().numpy()
            parts = [xyz[b, :int(cnt[b])] for b in range(len(cnt))]
            flat = torch.cat(parts).cpu().numpy() if len(parts) else np.zeros((0, 3), np.float32)
        else:
            side = self._side_stream(ctx)
            with torch.cuda.stream(side):
                side.wait_event(ready)
                xyz.record_stream(side)
                counts.record_stream(side)
                cnt = counts.cpu().numpy()
                parts = [xyz[b, :int(cnt[b])] for b in range(len(cnt))]
                flat = torch.cat(parts).cpu().numpy() if len(parts) else np.zeros((0, 3), np.float32)
        out, p = [], 0
        for n in cnt:
            out.append(flat[p:p + int(n)].copy())
            p += int(n)
        return out

    # ------------------------------------------------------------------ weights / checkpoints
    def _transforms(self):
        raise NotImplementedError

    def init_weights(self, seed=None):
        """Seeded Glorot-uniform kernels / zero biases (Keras defaults) and tfc default entropy models:
        what tf.global_variables_initializer() gives before saver.restore (compress_octree.py:83-92)."""
        rng = np.random.default_rng(self.seed if seed is None else seed)
        for prefix, tr, cin in self._transforms():
            MT.init_transform(tr, cin, rng)
        self._init_entropy(None)
        self._drop_device_state()

    def _drop_device_state(self):
        """Everything on the GPU that was derived from the weights / entropy tables (new ones have just been set): the cached
        medians / scale table tensors and the pcc_codec_desc, which holds raw device pointers to them."""
        self._dev_cache.clear()
        self._codec_cache = (None,)
        self.__dict__.pop('_stagings', None)

    def get_weights(self):
        out = {}
        for prefix, tr, _ in self._transforms():
            out.update(MT.get_weights(tr, prefix))
        out.update(self._entropy_weights())
        return out

    def set_weights(self, params):
        for prefix, tr, _ in self._transforms():
            MT.set_weights(tr, prefix, params)
        self._init_entropy(params)
        self._drop_device_state()

    def save_checkpoint(self, checkpoint_dir):
        os.makedirs(checkpoint_dir, exist_ok=True)
        np.savez(os.path.join(checkpoint_dir, CHECKPOINT_FILE), **self.get_weights())

    def restore(self, checkpoint_dir):
        """tf.train.Saver().restore(sess, tf.train.latest_checkpoint(dir)) (compress_octree.py:82,90-92): `model.npz`, or
        -- when the directory holds the reference's own TF1 checkpoint -- its TensorBundle files (tf_checkpoint.py)."""
        path = os.path.join(checkpoint_dir, CHECKPOINT_FILE)
        if not os.path.exists(path) and os.path.isdir(checkpoint_dir):
            from . import tf_checkpoint
            prefix = tf_checkpoint.latest_checkpoint(checkpoint_dir)
            if prefix is not None and os.path.exists(prefix + '.index'):
                tf_checkpoint.import_checkpoint(checkpoint_dir, self)
                return
        assert os.path.exists(path), f'Checkpoint {checkpoint_dir} was not found'
        with np.load(path) as f:
            self.set_weights({k: f[k] for k in f.files})

    # ------------------------------------------------------------------ block loops
    def encode_block_range(self, sess, blocks, resolution, with_normals=False, opt_metrics=('d1_mse',),
                           max_deltas=(np.inf,), fixed_threshold=False, debug=False):
        """The per-block part of compress_blocks (model_types.py:192-212) for a list of blocks: returns
        (strings per block, best-threshold list per block, candidate point lists per block, metric names,
        debug).  This is the unit that shards across GPUs (sharding.py)."""
        ctx = self._ctx(sess)
        dhw = self._spatial(self.x_shape)
        if not fixed_threshold:
            # once per call what the reference asserts per block (model_opt.py:22-24): unknown metric names, d2_* without normals
            from .utils.pc_metric import validate_opt_metrics
            validate_opt_metrics(opt_metrics, with_normals)
        strings_list, threshold_list, debug_t_list, x_hat_list = [], [], [], []
        opt_metrics_ret = metric_names(opt_metrics, max_deltas)
        half = len(self.thresholds) // 2
        SEARCH_LAG = 8            # chunks whose x_hat (batch x 1 MiB) stays on the GPU while their host jobs are in the pool
        pending = []

        def finalize_search(item):
            """decisions + candidate point lists of one chunk whose tallies (GPU) / host results are complete"""
            nonlocal opt_metrics_ret
            chunk_, x_hat_ = item['chunk'], item['x_hat']
            n_m = len(max_deltas) * len(opt_metrics)
            host = [f.result() for f in item['futures']] if item['futures'] is not None else None
            if host is not None and host and isinstance(host[0][1], tuple):      # 'tally_pruned' jobs: (tallies, (mean_tally, thresholds evaluated exactly))
                self.search_trees_built = getattr(self, 'search_trees_built', 0) + sum(h[1][1] for h in host)
                self.search_trees_total = getattr(self, 'search_trees_total', 0) + sum(len(h[0]) for h in host)
                host = [(h[0], h[1][0]) for h in host]
            if item['d1'] is not None:
                opt_metrics_ret, best_all = decide_from_tallies(chunk_, item['d1'], len(self.thresholds), resolution, opt_metrics, max_deltas, host,
                                                                gpu_d2=item['gpu_d2'])
            else:
                opt_metrics_ret, best_all = host[0][0], [bt for _, bt in host]
            # a block whose decode is empty at every threshold returns len(opt_metrics) entries (model_opt.py:35-36); with
            # more than one max_delta the reference's zip(*...) would silently drop the other candidates of the WHOLE
            # cloud -- here the 'emit nothing' index is repeated instead
            best_all = [list(bt) + [bt[-1]] * (n_m - len(bt)) for bt in best_all]
            per_metric = []
            for m in range(n_m):
                xyz, counts = self._extract_points(ctx, x_hat_, [bt[m] for bt in best_all], clip=True)
                per_metric.append(self._gather_points(xyz, counts))
            for j in range(len(chunk_)):
                threshold_list.append(list(best_all[j]))
                x_hat_list.append([per_metric[m][j] for m in range(n_m)])

        for c0 in range(0, len(blocks), self.batch_size):
            chunk = blocks[c0:c0 + self.batch_size]
            x = self._voxelize(ctx, chunk, dhw)
            enc = self._encode_batch(ctx, x, debug, thr=self._thr_tensor(ctx, [half] * len(chunk)) if fixed_threshold else None)
            x_hat = enc['x_hat']
            if fixed_threshold:
                # compute_optimal_thresholds' fixed branch (model_opt.py:27-31): index len//2 for every metric
                n_m = len(max_deltas) * len(opt_metrics)
                xyz, counts = enc['xyz'], enc['counts']
                strings = enc['finish']()
                pts = self._gather_points(xyz, counts)
                for j in range(len(chunk)):
                    threshold_list.append([half] * n_m)
                    x_hat_list.append([pts[j]] * n_m)
            else:
                # adaptive search (model_opt.py:33-73).  d1_* metrics: exact distance transforms on the GPU, with or without
                # normals in the input.  d2_* metrics: the reference's numbers depend on WHICH of several equidistant nearest
                # neighbours scipy's KD-tree returns (measured: another tie rule moves d2_mse by up to 60 % and the chosen
                # threshold in 2 of 6 blocks), so those tallies come from the same KD-trees on the host -- one block per
                # worker process of a persistent pool (the reference runs the blocks one after the other).  The host jobs of a
                # chunk are only QUEUED here; the GPU goes on with the next chunks and the decisions are taken (on the merged
                # table) a few chunks later, so the pool always holds several chunks' worth of blocks.
                want_d2 = any(m.startswith('d2_') for m in opt_metrics)
                on_gpu = gpu_search_supported(opt_metrics, dhw)
                gpu_d2 = want_d2 and on_gpu and d2_on_gpu(getattr(self, 'd2_search', None))          # nearest-index transforms, stated tie rule: opt-in (DESIGN_HISTORY.md 3.8)
                strings = enc['finish']()
                item = dict(chunk=chunk, x_hat=x_hat, futures=None, d1=None, gpu_d2=gpu_d2)
                # (round 6) the host pool only builds the A->B trees of the thresholds that can still win a d2 metric: the workers get the
                # GPU's exact D1 tallies as bounds (model_opt.host_threshold_stats_pruned); PCC_D2_NO_PRUNE=1: every threshold (A/B)
                prune = on_gpu and want_d2 and not gpu_d2 and not os.environ.get('PCC_D2_NO_PRUNE')
                if prune:
                    item['d1'] = d1_tallies_gpu(ctx, chunk, x_hat, self.thresholds)
                if (want_d2 and not gpu_d2) or not on_gpu:
                    xh = np.clip(x_hat.cpu().numpy(), 0.0, 1.0)
                    # blocks go over in their own dtype: the worker computes exactly what the in-process call would
                    if prune:
                        jobs = [('tally_pruned', np.ascontiguousarray(chunk[j]), xh[j], self.thresholds, with_normals, item['d1'][j], resolution,
                                 list(opt_metrics), list(max_deltas)) for j in range(len(chunk))]
                    elif on_gpu:
                        jobs = [('tally', np.ascontiguousarray(chunk[j]), xh[j], self.thresholds, with_normals) for j in range(len(chunk))]
                    else:
                        jobs = [('decide', np.ascontiguousarray(chunk[j]), xh[j], self.thresholds, resolution, with_normals,
                                 list(opt_metrics), list(max_deltas)) for j in range(len(chunk))]
                    self.host_search_jobs = getattr(self, 'host_search_jobs', 0) + len(jobs)
                    self.last_host_job_kind = jobs[0][0] if jobs else None
                    pool = self._search_pool(len(blocks))
                    item['futures'] = [pool.submit(job) for job in jobs]
                if on_gpu and item['d1'] is None:
                    item['d1'] = (d12_tallies_gpu if gpu_d2 else d1_tallies_gpu)(ctx, chunk, x_hat, self.thresholds)
                pending.append(item)
                if len(pending) > SEARCH_LAG:
                    finalize_search(pending.pop(0))
            strings_list.extend(strings)
            debug_t_list.extend(enc['debug'])
        while pending:
            finalize_search(pending.pop(0))
        return strings_list, threshold_list, x_hat_list, opt_metrics_ret, debug_t_list

    def compress_blocks(self, sess, blocks, binstr, points, resolution, level, with_normals=False,
                        opt_metrics=('d1_mse',), max_deltas=(np.inf,), fixed_threshold=False, debug=False,
                        need_points=True):
        """Uses the compression model to compress a point cloud (model_types.py:184-218).  Under torch.distributed (one
        process per GPU) the block list is sharded (sharding.py): rank 0 returns the complete result, the other ranks
        return (None, metadata without point lists, local debug list).  `need_points=False` skips the gather of the
        decoded candidate point lists to rank 0 (they are only needed for --dec_files / --debug)."""
        from . import sharding
        rank, world = sharding.world_info()
        # The KD-tree over the ORIGINAL cloud (the whole-cloud metrics of every candidate query it: pc_metric.py:80) does not depend on the
        # encode: it is built on a helper thread while the GPU codes the blocks (0.11 s of a 614 k-point cloud's 0.30 s; scipy builds
        # without the GIL).  Same constructor call as before: the same tree, the same neighbour picks.
        self._tree_future = self._helper_thread('tree').submit(cKDTree, points[:, :3]) if len(points) else None
        if world == 1:
            try:
                strings_list, threshold_list, x_hat_list, opt_metrics_ret, debug_t_list = self.encode_block_range(
                    sess, blocks, resolution, with_normals, opt_metrics, max_deltas, fixed_threshold, debug)
            finally:
                tree = self._tree_future.result() if self._tree_future is not None else None
                self._tree_future = None
            # block -> opt metric to opt metric -> block
            threshold_list = list(zip(*threshold_list))
            x_hat_list = list(zip(*x_hat_list))
            metadata = select_best_per_opt_metric(binstr, x_hat_list, level, opt_metrics_ret, points, resolution, with_normals, tree=tree)
            data_list = [list(zip(strings_list, threshold_list[x['idx']])) for x in metadata]
            return data_list, metadata, debug_t_list
        return self._compress_blocks_sharded(sess, blocks, binstr, points, resolution, level, with_normals, opt_metrics,
                                             max_deltas, fixed_threshold, debug, need_points)

    def _compress_blocks_sharded(self, sess, blocks, binstr, points, resolution, level, with_normals, opt_metrics,
                                 max_deltas, fixed_threshold, debug, need_points):
        from . import sharding
        from .utils.octree_coding import block_origins
        rank, world = sharding.world_info()
        lo, hi = sharding.shard_range(len(blocks), rank, world)
        try:
            strings_l, thr_l, xhat_l, names, debug_t_list = self.encode_block_range(
                sess, blocks[lo:hi], resolution, with_normals, opt_metrics, max_deltas, fixed_threshold, debug)
        finally:
            fut, self._tree_future = getattr(self, '_tree_future', None), None
            tree_a = fut.result() if fut is not None else None
        if tree_a is None:
            tree_a = cKDTree(points[:, :3])
        n_str = 1 if isinstance(self, CompressionModelV1) else 2
        n_m = len(max_deltas) * len(opt_metrics)
        if names is None or not len(blocks[lo:hi]):
            names = metric_names(opt_metrics, max_deltas)
        # What crosses ranks (sharding.py).  Per block one int64 row: string lengths, threshold index and candidate point count per metric.
        # The D1/D2 numbers of every candidate (select_best_per_opt_metric, src/model_types.py:128-176) come from per-rank partial tallies
        # of the pairs a rank OWNS (its decoded point is the nearest one to an original point: MIN over ranks of `d2 * world + rank`).
        origins = block_origins(binstr, [0, 0, 0], [resolution] * 3, level)[lo:hi]
        p1, p1_n = points[:, :3], get_normals_if(points, with_normals)
        cand_global = []
        for m in range(n_m):
            parts = [np.asarray(xhat_l[j][m], np.float64).reshape(-1, 3) + np.asarray(origins[j], np.float64) for j in range(hi - lo)]
            cand_global.append(np.vstack(parts) if parts else np.zeros((0, 3)))
        width = n_str + 2 * n_m
        rows = np.zeros((hi - lo, width), np.int64)
        for j in range(hi - lo):
            rows[j, :n_str] = [len(x) for x in strings_l[j]]
            rows[j, n_str:n_str + n_m] = thr_l[j][:n_m]
            rows[j, n_str + n_m:] = [len(x) for x in xhat_l[j][:n_m]]
        per_rank = sharding.shard_sizes(len(blocks), world)      # known to every rank: no size exchange anywhere below
        first = np.concatenate([[0], np.cumsum(per_rank)])
        my_strings = b''.join(x for ss in strings_l for x in ss)
        if 8 * len(p1) * n_m * world <= int(os.environ.get('PCC_KEY_GATHER_MAX_BYTES', 64 << 20)):
            # TWO collectives per cloud: (1) ONE all_gather of the rows with the MIN keys of all candidates riding as extra
            # rows (sharding.PiggybackGroup: every rank takes the MIN itself), (2) ONE all_gather of bytes: strings + the partial tallies
            grp = sharding.PiggybackGroup(rows, per_rank)
            part_tallies, have = cloud_metrics_batch(p1, cand_global, resolution - 1, p1_n, tree_a, grp, partial=True)
            table = grp.table
            tb = np.ascontiguousarray(part_tallies, np.float64).tobytes()
            payloads = sharding.all_gather_bytes(my_strings + tb, counts=[int(table[first[r]:first[r + 1], :n_str].sum()) + len(tb) for r in range(world)])
            blobs = [p[:len(p) - len(tb)] for p in payloads]
            tallies = np.zeros_like(part_tallies, dtype=np.float64)
            for p in payloads:          # rank order: every rank gets the same doubles
                tallies += np.frombuffer(p[len(p) - len(tb):], np.float64).reshape(part_tallies.shape)
        else:
            # a cloud whose keys (8 B per original point and candidate) are too many to move `world` times: THREE collectives -- (1) ONE
            # all_reduce(MIN) of the keys, (2) ONE all_gather of the rows + T rows with the bit patterns of the partial tallies (summed in
            # rank order), (3) ONE padded uint8 gather of the strings to rank 0
            part_tallies, have = cloud_metrics_batch(p1, cand_global, resolution - 1, p1_n, tree_a, sharding.RankGroup(), partial=True)
            T = -(-part_tallies.size // width)
            send = np.zeros((hi - lo + T, width), np.int64)
            send[:hi - lo] = rows
            send[hi - lo:].reshape(-1)[:part_tallies.size] = np.ascontiguousarray(part_tallies, np.float64).reshape(-1).view(np.int64)
            gathered = sharding.all_gather_rows(send, counts=[n + T for n in per_rank])
            ends = np.cumsum([n + T for n in per_rank])
            table = np.concatenate([gathered[e - n - T:e - T] for e, n in zip(ends, per_rank)], 0)
            tallies = np.zeros_like(part_tallies, dtype=np.float64)
            for e in ends:
                tallies += gathered[e - T:e].reshape(-1)[:part_tallies.size].view(np.float64).reshape(part_tallies.shape)
            blobs = sharding.gather_bytes(my_strings, counts=[int(table[first[r]:first[r + 1], :n_str].sum()) for r in range(world)])
        assert table.shape[0] == len(blocks)
        # the selection is replicated: every rank holds the summed tallies
        cand_metrics = finish_metrics(len(p1), tallies, have, resolution - 1, p1_n is not None)
        metadata = [{'idx': m, 'metrics': met} for _, m, met in rank_candidates(names, cand_metrics)]
        # (4) the reconstruction of the selected candidates on rank 0 (only for --dec_files / --debug)
        if need_points:
            for md in metadata:
                m = md['idx']
                n_pts = table[:, n_str + n_m + m]
                flat = sharding.gather_rows(np.vstack([np.asarray(xhat_l[j][m], np.float32).reshape(-1, 3) for j in range(hi - lo)])
                                            if hi > lo else np.zeros((0, 3), np.float32),
                                            counts=[int(n_pts[first[r]:first[r + 1]].sum()) for r in range(world)])
                if rank == 0:
                    off = np.concatenate([[0], np.cumsum(n_pts)])
                    md['x_hat_list'] = tuple(flat[off[j]:off[j + 1]] for j in range(len(blocks)))
                    md['blocks_depart'] = departition_octree(md['x_hat_list'], binstr, [0, 0, 0], [resolution] * 3, level)
                    md['blocks_full'] = np.vstack(md['blocks_depart'])
        if rank != 0:
            return None, metadata, debug_t_list
        # rank 0: split the gathered strings back into per-block tuples, block order == rank order
        strings_list, raw, pos = [], b''.join(blobs), 0
        for j in range(len(blocks)):
            ss = []
            for k in range(n_str):
                ss.append(raw[pos:pos + int(table[j, k])])
                pos += int(table[j, k])
            strings_list.append(tuple(ss))
        assert pos == len(raw)
        data_list = [list(zip(strings_list, [int(t) for t in table[:, n_str + md['idx']]])) for md in metadata]
        return data_list, metadata, debug_t_list

    def roundtrip_stream(self, sess, dense_chunks, thr_idx=None, gather=True):
        """Streams chunks of dense occupancy grids (each (b,D,H,W) float32 on the GPU) through the compress
        graph and the decompress graph with fixed-threshold extraction on both sides -- the unit of work of
        SURVEY.md §8d -- as a 3-deep software pipeline:
            chunk k   : GPU analysis/hyper + synthesis(enc) | host range-encode + z-decode
            chunk k-1 : host y-decode | GPU synthesis(dec)
            chunk k-2 : decoded points gathered to the host
        Yields (strings, enc_counts, dec_points or dec_counts) per chunk, in order."""
        ctx = self._ctx(sess)
        thr_idx = len(self.thresholds) // 2 if thr_idx is None else thr_idx
        q_b, q_g = [], []

        def stage_b(item):
            strings, cnt_e, st, dhw, B, yfut = item
            kw = {} if yfut is None else {'host': yfut.result()}
            dec = self._decode_phase_b(ctx, st, dhw, False, thr=self._thr_tensor(ctx, [thr_idx] * B), **kw)
            xyz_d, cnt_d = dec['xyz'], dec['counts']
            ready = torch.cuda.Event()
            ready.record(torch.cuda.current_stream(ctx.device))
            item_g = (strings, cnt_e, xyz_d, cnt_d, ready)
            # (round 5) few long streams per chunk: the gather (counts to the host, pack, 10 MB of points per 8-block chunk of 128^3 blocks:
            # 0.9 ms of blocking copies) runs on a helper thread too; results are still yielded in order
            return self._helper_thread('gather').submit(stage_g_work, item_g) if threaded else item_g

        def stage_g(item):
            return item.result() if threaded else stage_g_work(item)

        def stage_g_work(item):
            strings, cnt_e, xyz_d, cnt_d, ready = item
            if threaded:
                torch.cuda.set_device(ctx.device)
            pts = self._gather_points(xyz_d, cnt_d, ctx, ready) if gather else None
            with torch.cuda.stream(self._side_stream(ctx)):     # never a blocking copy on the main stream
                self._side_stream(ctx).wait_event(ready)
                cnt_e.record_stream(self._side_stream(ctx))
                ce = cnt_e.cpu().numpy()
                if not gather:
                    cnt_d.record_stream(self._side_stream(ctx))
                    pts = cnt_d.cpu().numpy()
            return strings, ce, pts

        def stage_a(item):
            enc, dhw, B = item
            strings = enc['strings'].result()
            st = self._decode_phase_a(ctx, strings, dhw)
            # (round 5) the y range-decode of this chunk -- host work only -- starts now on a helper thread and is collected one iteration
            # later by stage_b, when both coders fit the usable cores side by side (the rule of the encoder's helper thread below)
            yfut = self._helper_thread('ydec').submit(self._decode_phase_b_host, st) if threaded and hasattr(self, '_decode_phase_b_host') else None
            q_b.append((strings, enc['counts'], st, dhw, B, yfut))
            if len(q_b) > 1:
                q_g.append(stage_b(q_b.pop(0)))

        # The GPU work of chunk k+1 (compress graph) is enqueued BEFORE the host waits for the symbols of chunk k: the device
        # queue then always holds at least one more compress graph than the host coder needs to stay ahead of, so a slow or
        # noisy host does not drain it.  Pinned symbol buffers are per slot (three chunks can be between enqueue and coding).
        # The range ENCODER of a chunk (wait for its symbols, code them) runs on a helper thread -- the coder is sequential per
        # stream, so at 128^3 one chunk's y streams cost the host milliseconds -- while this thread range-DECODES an older chunk
        # and feeds the GPU; the library calls release the GIL.
        q_a, k = [], 0
        threaded = None
        trace = os.environ.get('PCC_STAGE_TIMES')          # host time per pipeline stage and iteration (ms), to stderr
        try:
            for x in dense_chunks:
                B, dhw = x.shape[0], tuple(x.shape[1:4])
                if threaded is None:
                    threaded = (2 * B <= _usable_cores() or bool(os.environ.get('PCC_FORCE_HELPER_THREADS'))) and not os.environ.get('PCC_NO_HELPER_THREADS')
                t0 = time.perf_counter()
                enc = self._encode_batch(ctx, x, False, thr=self._thr_tensor(ctx, [thr_idx] * B), slot=k % 3)
                # (only when both coders fit the usable cores side by side: with 32 streams per call on a 16-core container the two
                # would just take turns, with scheduler jitter on top -- measured: 7-12 ms hiccups in the 64^3 headline)
                enc['strings'] = self._coder_thread().submit(enc['finish']) if 2 * B <= _usable_cores() else _Immediate(enc['finish'])
                k += 1
                q_a.append((enc, dhw, B))
                t1 = time.perf_counter()
                if len(q_a) > 1:
                    stage_a(q_a.pop(0))
                t2 = time.perf_counter()
                if len(q_g) > 1:
                    out = stage_g(q_g.pop(0))
                    if trace:
                        print(f'stage ms: enqueue {1e3 * (t1 - t0):.2f} decode+enqueue {1e3 * (t2 - t1):.2f} gather {1e3 * (time.perf_counter() - t2):.2f}',
                              file=sys.stderr)
                    yield out
            while q_a:
                stage_a(q_a.pop(0))
            while q_b:
                q_g.append(stage_b(q_b.pop(0)))
            while q_g:
                yield stage_g(q_g.pop(0))
        finally:
            # an abandoned generator (or a stage that raised) must not leave helper-thread jobs running against buffers the caller is
            # about to free: wait for whatever is still queued
            for enc_, _, _ in q_a:
                fut = enc_.get('strings')
                if hasattr(fut, 'cancel') and not fut.cancel():
                    try:
                        fut.result()
                    except Exception:
                        pass
            for item in list(q_b) + list(q_g):
                for fut in (item if isinstance(item, tuple) else (item,)):
                    if hasattr(fut, 'cancel') and hasattr(fut, 'result') and not fut.cancel():
                        try:
                            fut.result()
                        except Exception:
                            pass

    def decompress_blocks(self, sess, blocks, x_shape, debug=False):
        """Uses the decompression model to decompress a point cloud (model_types.py:220-238).
        Software pipeline over chunks: the host range decoder of chunk k overlaps the synthesis of k-1."""
        from . import sharding
        rank, world = sharding.world_info()
        if world > 1 and not getattr(self, '_in_shard', False):
            # contiguous shards; the decoded float32 points go to rank 0 with one (counts, rows) gather -- the other ranks
            # return None (rank 0 writes the file, decompress_octree.py:111-113)
            lo, hi = sharding.shard_range(len(blocks), rank, world)
            self._in_shard = True
            try:
                local, dbg = self.decompress_blocks(sess, blocks[lo:hi], x_shape, debug)
            finally:
                self._in_shard = False
            per_rank = sharding.shard_sizes(len(blocks), world)
            first = np.concatenate([[0], np.cumsum(per_rank)])
            counts = sharding.all_gather_rows(np.array([[len(b)] for b in local], np.int64).reshape(-1, 1), counts=per_rank)[:, 0]
            flat = sharding.gather_rows(np.vstack(local).astype(np.float32) if len(local) else np.zeros((0, 3), np.float32),
                                        counts=[int(counts[first[r]:first[r + 1]].sum()) for r in range(world)])
            if rank != 0:
                return None, dbg
            off = np.concatenate([[0], np.cumsum(counts)])
            return [flat[off[j]:off[j + 1]] for j in range(len(blocks))], dbg
        ctx = self._ctx(sess)
        dhw = self._spatial(x_shape)
        chunks = [blocks[c0:c0 + self.batch_size] for c0 in range(0, len(blocks), self.batch_size)]
        state = [None] * len(chunks)
        results = [None] * len(chunks)
        for k in range(len(chunks) + 1):
            if k < len(chunks):
                state[k] = self._decode_phase_a(ctx, [s for s, _ in chunks[k]], dhw)
            if k >= 1:
                thr_idx = [int(t) for _, t in chunks[k - 1]]
                dec = self._decode_phase_b(ctx, state[k - 1], dhw, debug, thr=self._thr_tensor(ctx, thr_idx))
                xyz, counts = dec['xyz'], dec['counts']                 # the decoder does not clip (:232-233)
                results[k - 1] = (xyz, counts, dec['debug'])
                state[k - 1] = None
        dec_blocks, debug_t_list = [], []
        for xyz, counts, dbg in results:
            dec_blocks.extend(self._gather_points(xyz, counts))
            debug_t_list.extend(dbg)
        return dec_blocks, debug_t_list


def _np(t):
    return t.detach().cpu().numpy()


class CompressionModelV1(CompressionModel):
    def __init__(self, num_filters=32,
                 analysis_transform_type=TransformType.AnalysisTransformV1,
                 synthesis_transform_type=TransformType.SynthesisTransformV1, *args, **kwargs):
        self.num_filters = num_filters
        self._bind_transforms(analysis=analysis_transform_type, synthesis=synthesis_transform_type)
        self.entropy_bottleneck = None
        super().__init__(*args, **kwargs)

    def train(self, x, gamma, alpha, lmbda):
        raise NotImplementedError('training is out of scope of the MI355X hot path (SURVEY.md §2); '
                                  'the focal-loss reduction is available as utils.focal_loss.focal_loss')

    def _transforms(self):
        t = []
        if self.analysis_transform is not None:
            t.append(('analysis', self.analysis_transform, 1))
        t.append(('synthesis', self.synthesis_transform, self.num_filters))
        return t

    def _init_entropy(self, params):
        eb_params = tables = None
        if params is not None and 'entropy_bottleneck/quantiles' in params:
            eb_params = {k.split('/', 1)[1]: params[k] for k in params if k.startswith('entropy_bottleneck/')}
            if 'quantized_cdf' in eb_params:
                tables = (eb_params.pop('quantized_cdf'), eb_params.pop('cdf_length'), eb_params.pop('offset'))
        self.entropy_bottleneck = EntropyBottleneck(self.num_filters, params=eb_params, tables=tables, seed=self.seed)

    def _entropy_weights(self):
        eb = self.entropy_bottleneck
        out = {f'entropy_bottleneck/{k}': v for k, v in eb.params.items()}
        out.update({'entropy_bottleneck/quantized_cdf': eb.quantized_cdf, 'entropy_bottleneck/cdf_length': eb.cdf_length,
                    'entropy_bottleneck/offset': eb.offset})
        return out

    def compress(self, x_shape):
        """Initializes the compression model (model_types.py:283-295)."""
        self.x_shape = [int(v) for v in x_shape]
        self.analysis_transform = self.analysis_transform_class(self.num_filters, data_format=self.data_format)
        self.synthesis_transform = self.synthesis_transform_class(self.num_filters, data_format=self.data_format)
        self.init_weights()

    def decompress(self):
        """Initializes the decompression model (model_types.py:297-309)."""
        self.analysis_transform = None
        self.synthesis_transform = self.synthesis_transform_class(self.num_filters, data_format=self.data_format)
        self.init_weights()

    # ---- batched graph
    def _encode_batch(self, ctx, x, debug, thr=None, slot=0):
        B = x.shape[0]
        eb = self.entropy_bottleneck
        codec = self._codec(ctx)
        t = {}
        ready = None
        stg = self._staging(ctx, slot, B, [v // 8 for v in x.shape[1:4]])
        if codec is not None:                      # analysis -> quantise -> synthesis (-> fixed threshold) in one ABI call
            ready = torch.cuda.Event()
            ready.record(self._side_stream(ctx, '_idx_stream'))      # creates the handle (on a stream that is idle now: a record costs the
                                                                     # main queue a few microseconds); re-recorded by the library on the main stream
            t = ops.codec_encode(ctx, codec, x.contiguous(), thr, symbols_ready=ready, staging=stg)
            y, ysym, y_hat, x_hat = t['y'], t['symbols'], t['y_hat'], t['x_hat']
        else:
            med = self._dev(ctx, 'medians', eb.medians)
            y = self.analysis_transform.forward_ndhwc(ctx, x.unsqueeze(-1))
            ysym, y_hat = ops.quantize(ctx, y, med, self.round_mode)
            stg.pack(ctx, ysym)
        ev = self._ship(ctx, stg, ready)
        if codec is None:
            x_hat = self.synthesis_transform.forward_ndhwc(ctx, y_hat)[..., 0].contiguous()
            if thr is not None:
                t['xyz'], t['counts'] = ops.threshold_compact(ctx, x_hat, thr, clip=True)
        rows, mod = self._eb_rows(ysym[0].numel(), self.num_filters)

        def finish():
            ev.synchronize()
            # a symbol beyond the narrow host type (never seen in practice) shows in the tile maxima: fetch that tensor as int32
            ys_src = stg.ysym if stg.sym_dtype == torch.int32 or int(stg.ytm.max()) <= 32767 else self._to_stream_order(ysym).cpu()
            ys = ops.range_encode_batch(eb.table, ys_src.view(B, -1), rows, mod, self.coder_threads)
            return [(s,) for s in ys]

        dbg = [{'y': _np(y[b:b + 1]), 'symbols': _np(ysym[b:b + 1]), 'y_hat': _np(y_hat[b:b + 1]),
                'x_hat': _np(x_hat[b:b + 1].unsqueeze(-1))} for b in range(B)] if debug else [None] * B
        return dict(x_hat=x_hat, finish=finish, debug=dbg, xyz=t.get('xyz'), counts=t.get('counts'))

    def _decode_phase_a(self, ctx, strings, dhw):
        B = len(strings)
        eb = self.entropy_bottleneck
        yshape = self._stream_shape(B, [v // 8 for v in dhw], self.num_filters)
        ysym_h, ysym_release = self._pinned.ring('dec_ysym', yshape, _host_dtypes()[0])
        n = int(np.prod(yshape[1:]))
        rows, mod = self._eb_rows(n, self.num_filters)
        return dict(ysym=self._range_decode(eb.table, [s[0] for s in strings], n, rows, mod, ysym_h),
                    ysym_release=ysym_release)

    def _decode_phase_b(self, ctx, st, dhw, debug, thr=None):
        eb = self.entropy_bottleneck
        packed = self._symbols_to_device(ctx, st['ysym'], st['ysym_release'])
        codec = self._codec(ctx)
        if codec is not None:                      # unpack -> dequantise -> synthesis (-> threshold + compaction) in one ABI call
            t = ops.codec_decode_main(ctx, codec, None, dhw, thr, packed=packed, channels_first=self.data_format == 'channels_first')
            y_hat, x_hat = t['y_hat'], t['x_hat']
        else:
            ysym = self._unpack(ctx, packed, [v // 8 for v in dhw])
            y_hat = ops.dequantize(ctx, ysym, self._dev(ctx, 'medians', eb.medians))
            x_hat = self.synthesis_transform.forward_ndhwc(ctx, y_hat)[..., 0].contiguous()
            t = {}
            if thr is not None:
                t['xyz'], t['counts'] = ops.threshold_compact(ctx, x_hat, thr, clip=False)
        B = x_hat.shape[0]
        dbg = [{'y_hat': _np(y_hat[b:b + 1]), 'x_hat': _np(x_hat[b:b + 1].unsqueeze(-1))} for b in range(B)] if debug else [None] * B
        return dict(x_hat=x_hat, debug=dbg, xyz=t.get('xyz'), counts=t.get('counts'))


class CompressionModelV2(CompressionModel):
    def __init__(self, num_filters=32,
                 analysis_transform_type=TransformType.AnalysisTransformV1,
                 synthesis_transform_type=TransformType.SynthesisTransformV1,
                 hyper_analysis_transform_type=TransformType.HyperAnalysisTransform,
                 hyper_synthesis_transform_type=TransformType.HyperSynthesisTransform,
                 scales_min=0.11, scales_max=256, scales_levels=64, *args, **kwargs):
        self.num_filters = num_filters
        self._bind_transforms(analysis=analysis_transform_type, synthesis=synthesis_transform_type,
                              hyper_analysis=hyper_analysis_transform_type, hyper_synthesis=hyper_synthesis_transform_type)
        self.scale_table = scale_table(scales_min, scales_max, scales_levels)
        self.entropy_bottleneck = self.conditional_bottleneck = None
        super().__init__(*args, **kwargs)

    def train(self, x, gamma, alpha, lmbda):
        raise NotImplementedError('training is out of scope of the MI355X hot path (SURVEY.md §2); '
                                  'the focal-loss reduction is available as utils.focal_loss.focal_loss')

    def _transforms(self):
        t = []
        if self.analysis_transform is not None:
            t.append(('analysis', self.analysis_transform, 1))
            t.append(('hyper_analysis', self.hyper_analysis_transform, self.num_filters))
        t.append(('hyper_synthesis', self.hyper_synthesis_transform, self.num_filters))
        t.append(('synthesis', self.synthesis_transform, self.num_filters))
        return t

    def _init_entropy(self, params):
        CompressionModelV1._init_entropy(self, params)
        tables = None
        if params is not None and 'gaussian_conditional/quantized_cdf' in params:
            tables = tuple(params[f'gaussian_conditional/{k}'] for k in ('quantized_cdf', 'cdf_length', 'offset'))
        if self.conditional_bottleneck is None or tables is not None:
            self.conditional_bottleneck = GaussianConditional(self.scale_table, tables=tables)

    def _entropy_weights(self):
        out = CompressionModelV1._entropy_weights(self)
        gc = self.conditional_bottleneck
        out.update({'gaussian_conditional/quantized_cdf': gc.quantized_cdf, 'gaussian_conditional/cdf_length': gc.cdf_length,
                    'gaussian_conditional/offset': gc.offset})
        return out

    def compress(self, x_shape):
        """Initializes the compression model (model_types.py:371-391)."""
        self.x_shape = [int(v) for v in x_shape]
        F, df = self.num_filters, self.data_format
        self.analysis_transform = self.analysis_transform_class(F, data_format=df)
        self.synthesis_transform = self.synthesis_transform_class(F, data_format=df)
        self.hyper_analysis_transform = self.hyper_analysis_transform_class(F, data_format=df)
        self.hyper_synthesis_transform = self.hyper_synthesis_transform_class(F, data_format=df)
        self.init_weights()

    def decompress(self):
        """Initializes the decompression model (model_types.py:393-411)."""
        F, df = self.num_filters, self.data_format
        self.analysis_transform = self.hyper_analysis_transform = None
        self.synthesis_transform = self.synthesis_transform_class(F, data_format=df)
        self.hyper_synthesis_transform = self.hyper_synthesis_transform_class(F, data_format=df)
        self.init_weights()

    # ---- batched graph: x -A-> y -HA-> z -EB-> z_string ; z_hat -HS-> sigma ; (y, sigma) -GC-> y_string ; y_hat -S-> x_hat
    def _encode_batch(self, ctx, x, debug, thr=None, slot=0):
        B = x.shape[0]
        F = self.num_filters
        eb, gc = self.entropy_bottleneck, self.conditional_bottleneck
        codec = self._codec(ctx)
        t = {}
        ready = None
        stg = self._staging(ctx, slot, B, [v // 8 for v in x.shape[1:4]], [v // 16 for v in x.shape[1:4]])
        if codec is not None:                      # the whole GPU part of compress() (model_types.py:379-388) in one ABI call
            ready = torch.cuda.Event()
            ready.record(self._side_stream(ctx, '_idx_stream'))      # creates the handle (on a stream that is idle now: a record costs the
                                                                     # main queue a few microseconds); re-recorded by the library on the main stream
            t = ops.codec_encode(ctx, codec, x.contiguous(), thr, symbols_ready=ready, staging=stg)
            y, z, zsym, z_hat, sigma, idx, ysym, y_hat, x_hat = (t[k] for k in ('y', 'z', 'z_symbols', 'z_hat', 'sigma_hat',
                                                                                'indexes', 'symbols', 'y_hat', 'x_hat'))
        else:
            med = self._dev(ctx, 'medians', eb.medians)
            tab = self._dev(ctx, 'scale_table', gc.scale_table_f32)
            y = self.analysis_transform.forward_ndhwc(ctx, x.unsqueeze(-1))
            z = self.hyper_analysis_transform.forward_ndhwc(ctx, y)
            zsym, z_hat = ops.quantize(ctx, z, med, self.round_mode)
            sigma = self.hyper_synthesis_transform.forward_ndhwc(ctx, z_hat)
            idx = ops.scale_to_index(ctx, sigma, tab)
            ysym, y_hat = ops.quantize(ctx, y, None, self.round_mode)
            stg.pack(ctx, ysym, zsym, idx)
        # what crosses PCIe: ONE buffer per chunk -- symbols as int16, the 64 scale rows as uint8 (8.5 -> 3.3 MB per 32-block
        # chunk), packed in stream order by the library before `ready`, plus the per-tile max|symbol| that tells the host
        # afterwards whether a symbol exceeded int16 (then that tensor is fetched again as int32: never seen in practice).
        # The copy runs on a side stream so that it overlaps the synthesis transform.
        ev = self._ship(ctx, stg, ready)
        if codec is None:
            x_hat = self.synthesis_transform.forward_ndhwc(ctx, y_hat)[..., 0].contiguous()
            if thr is not None:
                t['xyz'], t['counts'] = ops.threshold_compact(ctx, x_hat, thr, clip=True)
        rows, mod = self._eb_rows(zsym[0].numel(), F)

        def finish():
            ev.synchronize()  # symbols are on the host
            fits = stg.sym_dtype == torch.int32
            zs_src = stg.zsym if fits or int(stg.ztm.max()) <= 32767 else self._to_stream_order(zsym).cpu()
            ys_src = stg.ysym if fits or int(stg.ytm.max()) <= 32767 else self._to_stream_order(ysym).cpu()
            zs = ops.range_encode_batch(eb.table, zs_src.view(B, -1), rows, mod, self.coder_threads)
            ys = ops.range_encode_batch(gc.table, ys_src.view(B, -1), stg.idx.view(B, -1), 0, self.coder_threads)
            return list(zip(ys, zs))  # strings = (y_string, z_string), model_types.py:389

        dbg = [None] * B
        if debug:
            dbg = [{'y': _np(y[b:b + 1]), 'z': _np(z[b:b + 1]), 'z_symbols': _np(zsym[b:b + 1]),
                    'z_hat': _np(z_hat[b:b + 1]), 'sigma_hat': _np(sigma[b:b + 1]), 'indexes': _np(idx[b:b + 1]),
                    'symbols': _np(ysym[b:b + 1]), 'y_hat': _np(y_hat[b:b + 1]), 'x_hat': _np(x_hat[b:b + 1].unsqueeze(-1))}
                   for b in range(B)]
        return dict(x_hat=x_hat, finish=finish, debug=dbg, xyz=t.get('xyz'), counts=t.get('counts'))

    def _decode_phase_a(self, ctx, strings, dhw):
        """z_string -EB.decompress-> z_hat -HS-> sigma -> indexes (async D2H)."""
        B, F = len(strings), self.num_filters
        eb, gc = self.entropy_bottleneck, self.conditional_bottleneck
        zshape = self._stream_shape(B, [v // 16 for v in dhw], F)
        # per-slot cached pinned buffers (like the encoder's): nothing is allocated in the steady state
        zsym_h, zsym_release = self._pinned.ring('dec_zsym', zshape, _host_dtypes()[0])
        nz = int(np.prod(zshape[1:]))
        rows, mod = self._eb_rows(nz, F)
        zpacked = self._symbols_to_device(ctx, self._range_decode(eb.table, [s[1] for s in strings], nz, rows, mod, zsym_h), zsym_release)
        # the 64 scale rows leave in stream order, one byte each
        row_t = _host_dtypes(len(self.scale_table))[1]
        cf = self.data_format == 'channels_first'
        idx_s = torch.empty(self._stream_shape(B, [v // 8 for v in dhw], F), dtype=row_t, device=ctx.device)
        codec = self._codec(ctx)
        if codec is not None:                      # unpack z -> dequantise -> hyper-synthesis -> indexes -> pack, one ABI call
            t = ops.codec_decode_hyper(ctx, codec, None, dhw, packed=zpacked, channels_first=cf, idx_packed=idx_s)
            z_hat, sigma, idx = t['z_hat'], t['sigma_hat'], t['indexes']
        else:
            zsym = self._unpack(ctx, zpacked, [v // 16 for v in dhw])
            z_hat = ops.dequantize(ctx, zsym, self._dev(ctx, 'medians', eb.medians))
            sigma = self.hyper_synthesis_transform.forward_ndhwc(ctx, z_hat)
            idx = ops.scale_to_index(ctx, sigma, self._dev(ctx, 'scale_table', gc.scale_table_f32))
            ops.symbols_pack(ctx, idx, cf, idx_s.data_ptr(), idx_s.element_size())
        # (the host reads idx_h synchronously in phase b, long before the ring comes round: no release event needed)
        idx_h, _ = self._pinned.ring('dec_idx', idx_s.shape, row_t)
        packed = torch.cuda.Event()
        packed.record(torch.cuda.current_stream(ctx.device))
        side = self._side_stream(ctx, '_idx_stream')
        with torch.cuda.stream(side):               # (not on the main stream: the copy would delay the kernels queued behind it)
            side.wait_event(packed)
            idx_s.record_stream(side)
            idx_h.copy_(idx_s, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(side)
        return dict(strings=strings, idx_h=idx_h, ev=ev, z_hat=z_hat, sigma=sigma, idx=idx, zsym_h=zsym_h, device=ctx.device)

    def _decode_phase_b_host(self, st):
        """The host part of phase b: wait for the CDF-row indexes, range-decode the y strings into a pinned buffer.  No GPU work is
        enqueued here, so roundtrip_stream may run it on a helper thread (the coder is sequential per stream: 2 ms per 8-block chunk
        of 128^3 blocks) while the calling thread keeps feeding the device."""
        gc = self.conditional_bottleneck
        strings, idx_h = st['strings'], st['idx_h']
        B = len(strings)
        # may run on the 'ydec' helper thread: pinned allocations below must be made with THIS model's device current (a helper thread
        # starts on device 0 and would touch or create a context there on a rank that owns another GPU)
        if st.get('device') is not None:
            torch.cuda.set_device(st['device'])
        st['ev'].synchronize()
        ysym_h, ysym_release = self._pinned.ring('dec_ysym', idx_h.shape, _host_dtypes()[0])
        n = int(np.prod(idx_h.shape[1:]))
        return self._range_decode(gc.table, [s[0] for s in strings], n, idx_h.view(B, -1), 0, ysym_h), ysym_release

    def _decode_phase_b(self, ctx, st, dhw, debug, thr=None, host=None):
        """(y_string, indexes) -GC.decompress-> y_hat -S-> x_hat [-> thresholded points].  host: the result of _decode_phase_b_host when
        it already ran elsewhere."""
        strings = st['strings']
        B = len(strings)
        ysym, ysym_release = self._decode_phase_b_host(st) if host is None else host
        packed = self._symbols_to_device(ctx, ysym, ysym_release)
        codec = self._codec(ctx)
        if codec is not None:                      # unpack -> dequantise -> synthesis (-> threshold + compaction) in one ABI call
            t = ops.codec_decode_main(ctx, codec, None, dhw, thr, packed=packed, channels_first=self.data_format == 'channels_first')
            ysym, y_hat, x_hat = t['symbols'], t['y_hat'], t['x_hat']
        else:
            ysym = self._unpack(ctx, packed, [v // 8 for v in dhw])
            y_hat = ops.dequantize(ctx, ysym, None)
            x_hat = self.synthesis_transform.forward_ndhwc(ctx, y_hat)[..., 0].contiguous()
            t = {}
            if thr is not None:
                t['xyz'], t['counts'] = ops.threshold_compact(ctx, x_hat, thr, clip=False)
        dbg = [None] * B
        if debug:
            dbg = [{'z_hat': _np(st['z_hat'][b:b + 1]), 'sigma_hat': _np(st['sigma'][b:b + 1]),
                    'indexes': _np(st['idx'][b:b + 1]), 'symbols': _np(ysym[b:b + 1]), 'y_hat': _np(y_hat[b:b + 1]),
                    'x_hat': _np(x_hat[b:b + 1].unsqueeze(-1))} for b in range(B)]
        return dict(x_hat=x_hat, debug=dbg, xyz=t.get('xyz'), counts=t.get('counts'))


class ModelType(Enum):
    v1 = CompressionModelV1
    v2 = CompressionModelV2
