"""Octree block partitioning -- same results as /root/reference/src/utils/octree_coding.py:68-169,
re-designed as vectorised numpy (the reference loops over every point in Python, :106-108).

partition_octree : points -> (blocks in Morton order with x least significant, local coordinates,
                   original point order kept inside a block; binstr = pre-order occupancy bytes)
departition_octree: inverse; adds each block's origin back.
"""
import numpy as np


def _morton(ids, nbits):
    """ids (n,3) non-negative ints -> interleaved code, digit = z<<2 | y<<1 | x per level (MSB first)."""
    ids = ids.astype(np.uint64)
    code = np.zeros(len(ids), np.uint64)
    for b in range(nbits - 1, -1, -1):
        digit = (((ids[:, 2] >> np.uint64(b)) & np.uint64(1)) << np.uint64(2)) | \
                (((ids[:, 1] >> np.uint64(b)) & np.uint64(1)) << np.uint64(1)) | \
                ((ids[:, 0] >> np.uint64(b)) & np.uint64(1))
        code = (code << np.uint64(3)) | digit
    return code


def _binstr_from_codes(codes, level):
    """Pre-order occupancy bytes of the octree whose occupied leaves (depth `level`) are `codes`
    (sorted unique Morton codes).  Equals partition_octree_rec's binstr (octree_coding.py:50-61)."""
    nodes = []  # (first leaf code below the node, depth, occupancy byte)
    for d in range(level):
        shift = np.uint64(3 * (level - d))
        prefix = codes >> shift
        digit = ((codes >> np.uint64(3 * (level - d - 1))) & np.uint64(7)).astype(np.uint8)
        uniq, first = np.unique(prefix, return_index=True)
        occ = np.zeros(len(uniq), np.uint8)
        np.bitwise_or.at(occ, np.searchsorted(uniq, prefix), (np.uint8(1) << digit))
        nodes.append(np.stack([codes[first].astype(np.int64), np.full(len(uniq), d, np.int64), occ.astype(np.int64)], 1))
    nodes = np.concatenate(nodes)
    order = np.lexsort((nodes[:, 1], nodes[:, 0]))  # by first leaf, then depth: a parent precedes its children
    return [int(v) for v in nodes[order, 2]]


def _partition_native(points, block_size, level):
    """One counting sort in libpcc_geo_hip.so (csrc/octree.cpp, host code); None when the library is not built."""
    try:
        from .. import _lib as L
        lib = L.lib()
    except Exception:          # host-side helper: the numpy path below gives the same result
        return None
    n = len(points)
    order = np.empty(n, np.int64)
    counts = np.empty(8 ** level, np.int64)
    occ = lib.pcc_octree_bucket(points.ctypes.data, n, points.shape[1], block_size, level, order.ctypes.data, counts.ctypes.data)
    L.check(occ, 'pcc_octree_bucket')
    codes = np.flatnonzero(counts)
    ends = np.cumsum(counts[codes])
    starts = ends - counts[codes]
    # de-interleave the Morton codes of the occupied buckets back into block ids -> origins
    c = codes.astype(np.uint64)
    ids = np.zeros((len(codes), 3), np.int64)
    for b in range(level):
        for a in range(3):
            ids[:, a] |= (((c >> np.uint64(3 * b + a)) & np.uint64(1)).astype(np.int64) << b)
    sorted_pts = points[order]
    blocks = []
    for s_, e_, o in zip(starts, ends, ids * block_size):
        blk = sorted_pts[s_:e_].copy()
        blk[:, :3] -= o
        blocks.append(blk)
    return blocks, _binstr_from_codes(c, level)


def partition_octree(points, bbox_min, bbox_max, level):
    points = np.asarray(points)
    if len(points) == 0 or level == 0:
        return [points], None
    bbox_min = np.asarray(bbox_min)
    np.testing.assert_array_equal(bbox_min, [0, 0, 0])
    bbox_max = np.asarray(bbox_max)
    geo_level = int(np.ceil(np.log2(np.max(bbox_max))))
    assert geo_level >= level
    block_size = 2 ** (geo_level - level)

    if level <= 7 and points.dtype == np.float64 and points.flags['C_CONTIGUOUS'] and (block_size << level) < 2 ** 31:
        native = _partition_native(points, block_size, level)
        if native is not None:
            return native

    block_ids = (points[:, :3] // block_size).astype(np.uint32)
    # Morton key over the `level` bits of a block id.  The reference builds its key from
    # (geo_level - level)-wide binary strings (octree_coding.py:88-90), which is this order whenever
    # geo_level - level >= level (every configuration the reference runs) and is not injective otherwise;
    # the true Morton order is the one departition_octree needs, so it is used unconditionally.
    key = _morton(block_ids, level)
    order = np.argsort(key, kind='stable')  # stable: original point order is kept inside a block
    skey = key[order]
    starts = np.flatnonzero(np.concatenate([[True], skey[1:] != skey[:-1]]))
    ends = np.concatenate([starts[1:], [len(points)]])

    local = points.astype(np.float64, copy=True)
    local[:, :3] -= (block_ids.astype(np.int64) * block_size)
    local_sorted = local[order]
    blocks = [local_sorted[s:e] for s, e in zip(starts, ends)]
    binstr = _binstr_from_codes(skey[starts], level)
    return blocks, binstr


def compute_new_bbox(idx, bbox_min, bbox_max):
    midpoint = (bbox_max - bbox_min) // 2 + bbox_min
    cur_bbox_min = bbox_min.copy()
    cur_bbox_max = midpoint.copy()
    for a in range(3):
        if (idx >> a) & 1:
            cur_bbox_min[a] = midpoint[a]
            cur_bbox_max[a] = bbox_max[a]
    return cur_bbox_min, cur_bbox_max


def block_origins(binstr_list, bbox_min, bbox_max, level):
    """Origins of the occupied blocks in stream order (pre-order walk of the occupancy bytes)."""
    bbox_min = np.asarray(bbox_min).astype(np.int64)
    bbox_max = np.asarray(bbox_max).astype(np.int64)
    binstr = [int(b) for b in binstr_list]
    origins = []
    pos = 0
    stack = [(0, bbox_min, bbox_max)]
    # iterative pre-order: children must be visited in increasing index, so push them reversed
    while stack:
        depth, bmin, bmax = stack.pop()
        if depth == level:
            origins.append(bmin)
            continue
        byte = binstr[pos]
        pos += 1
        for i in range(7, -1, -1):
            if (byte >> i) & 1:
                cmin, cmax = compute_new_bbox(i, bmin, bmax)
                stack.append((depth + 1, cmin, cmax))
    assert pos == len(binstr), f'binstr not consumed completely ({pos}/{len(binstr)})'
    return origins


def departition_octree(blocks, binstr_list, bbox_min, bbox_max, level):
    origins = block_origins(binstr_list, bbox_min, bbox_max, level)
    assert len(origins) == len(blocks), f'{len(origins)} occupied leaves in binstr but {len(blocks)} blocks'
    out = []
    for b, o in zip(blocks, origins):
        b = np.asarray(b)
        out.append(b + np.pad(o, [0, b.shape[1] - 3]))
    return out
