"""Point-to-point (D1) and point-to-plane (D2) distortion between an original cloud A and a decoded cloud B.

Same numbers as /root/reference/src/utils/pc_metric.py:76-138 (pinned by tests/golden/model_opt*.npz and select_best.npz, which
the reference's own module produced), organised differently: the KD-trees are only asked for nearest-neighbour INDICES
(`nearest`), the distortion is reduced to a five-number tally per cloud pair (`pair_tally`), and every derived quantity
(sum / mse / psnr, per direction and symmetric) comes from one table builder (`metrics_table`) that also works on whole arrays
of tallies -- the threshold search (model_opt.py) evaluates all thresholds of a block with one call, the GPU search feeds it
exact integer sums, and the multi-GPU path (sharding.py) all-reduces the tally before building the table.  `compute_metrics`
is the single-process, one-candidate case of `cloud_metrics_batch`.

Conventions: `to_b[i]` = index (into B) of the nearest decoded point of original point i; `to_a[j]` = index (into A) of the
nearest original point of decoded point j.  Arithmetic runs in whatever dtype numpy promotes the operands to, like the
reference (float32 PLY points against float64 departitioned blocks -> float64).
"""
import numpy as np
from scipy.spatial import cKDTree

GROUPS = ('d1', 'd2')
_OPT_STEMS = ('sum_AB', 'sum_BA', 'sum_max', 'sum_mean', 'mse_AB', 'mse_BA', 'mse')
# psnr is not offered for optimisation: it is monotone in mse (pc_metric.py:55)
avail_opt_metrics = [f'{g}_{stem}' for g in GROUPS for stem in _OPT_STEMS]      # the reference's order (pc_metric.py:57-58): --help and assertion texts print it

# slots of a tally vector (float64[5]); additive over disjoint parts of B (B->A terms) and of A (A->B terms)
N_B, D1_AB, D1_BA, D2_AB, D2_BA = range(5)
_TALLY_SLOTS = {'d1': (D1_AB, D1_BA), 'd2': (D2_AB, D2_BA)}


def validate_opt_metrics(opt_metrics, with_normals=False):
    unknown = [m for m in opt_metrics if m not in avail_opt_metrics]
    assert not unknown, f'{unknown[0] if unknown else None} not found in {avail_opt_metrics}'
    needs_normals = [m for m in opt_metrics if m.split('_', 1)[0] == 'd2']
    assert with_normals or not needs_normals, f'{needs_normals[0] if needs_normals else None} not available without normals'


def psnr(mse, max_energy):
    with np.errstate(divide='ignore'):
        return 10 * np.log10(max_energy / mse)


def nearest(tree, queries):
    """Index of the nearest tree point for every query point.  Large clouds use all cores (the reference always does,
    pc_metric.py:80-81); for the small per-block queries of the threshold search the thread start-up dominates."""
    if len(queries) == 0:
        return np.zeros(0, np.int64)
    return tree.query(queries, workers=-1 if max(tree.n, len(queries)) > 200000 else 1)[1]


def point_gap(src, dst, link):
    """Residual vectors src[i] - dst[link[i]]."""
    return src - dst[link]


def squared_norms(gap):
    return (gap ** 2).sum(axis=1)


def plane_error(gap, normals):
    """Sum over points of the squared projection of the residual on the normal."""
    return (((gap * normals).sum(axis=1)) ** 2).sum()


def transfer_normals(a_normals, to_a, to_b, a_mask=None):
    """Normals for the decoded points (pc_metric.py:8-25): decoded point j gets the mean normal of the original points that
    have j as THEIR nearest decoded point (restricted to `a_mask` when given); a decoded point nobody points to takes the
    normal of its own nearest original point.  bincount accumulates in index order, i.e. in the order of the reference's loop."""
    n_b = len(to_a)
    src, link = (a_normals, to_b) if a_mask is None else (a_normals[a_mask], to_b[a_mask])
    votes = np.bincount(link, minlength=n_b).astype(np.float64)
    acc = np.stack([np.bincount(link, weights=src[:, c], minlength=n_b) for c in range(src.shape[1])], axis=1)
    orphan = votes == 0
    acc[orphan] += a_normals[to_a[orphan]]
    votes[orphan] = 1
    return acc / votes[:, None]


def pair_tally(a, b, to_b, to_a, a_normals=None, a_mask=None):
    """float64[5] tally (N_B, D1_AB, D1_BA, D2_AB, D2_BA) of the pair (A, B).  `a_mask` limits the A->B terms to a subset of
    the original points (the multi-GPU path: the points whose nearest decoded point lives on this rank)."""
    tally = np.zeros(5, np.float64)
    tally[N_B] = len(b)
    if len(b) == 0:
        return tally
    gap_ab, gap_ba = point_gap(a, b, to_b), point_gap(b, a, to_a)
    if a_mask is not None:
        gap_ab = gap_ab[a_mask]
    tally[D1_AB] = squared_norms(gap_ab).sum()
    tally[D1_BA] = squared_norms(gap_ba).sum()
    if a_normals is not None:
        b_normals = transfer_normals(a_normals, to_a, to_b, a_mask)
        tally[D2_AB] = plane_error(gap_ab, b_normals[to_b if a_mask is None else to_b[a_mask]])
        tally[D2_BA] = plane_error(gap_ba, a_normals[to_a])
    return tally


def metrics_table(n_a, tally, peak, groups=GROUPS):
    """The reference's metric dictionary (pc_metric.py:83-137) from tallies.  `tally` is float64[5] (-> scalars) or
    float64[T, 5] (-> arrays over T candidate clouds).  An empty B gives mse_BA = nan / psnr nan -- callers guard that."""
    tally = np.asarray(tally, np.float64)
    n_b = tally[..., N_B]
    energy = 3 * peak * peak
    out = {}
    with np.errstate(divide='ignore', invalid='ignore'):
        for g in groups:
            s_ab, s_ba = (tally[..., k] for k in _TALLY_SLOTS[g])
            per_dir = {'AB': (s_ab, s_ab / n_a), 'BA': (s_ba, s_ba / n_b)}
            out[f'{g}_sum_AB'], out[f'{g}_sum_BA'] = s_ab, s_ba
            out[f'{g}_sum_max'] = np.maximum(s_ab, s_ba)
            out[f'{g}_sum_mean'] = (s_ab + s_ba) / 2
            for d, (_, mse) in per_dir.items():
                out[f'{g}_mse_{d}'] = mse
            out[f'{g}_mse'] = np.maximum(per_dir['AB'][1], per_dir['BA'][1])
            for d, (_, mse) in per_dir.items():
                out[f'{g}_psnr_{d}'] = psnr(mse, energy)
            out[f'{g}_psnr'] = np.minimum(out[f'{g}_psnr_AB'], out[f'{g}_psnr_BA'])
    return out


class SingleProcess:
    """The world-size-1 communicator of `cloud_metrics_batch` (sharding.RankGroup is the torch.distributed one)."""
    rank, world = 0, 1

    def claim(self, sq_dist_ab, have_points):
        """Per candidate cloud: which original points have their globally nearest decoded point on this rank -- all of them
        here (None = no decoded point anywhere)."""
        return [np.ones(len(d), bool) if h else None for d, h in zip(sq_dist_ab, have_points)]

    def total(self, tallies):
        return tallies


def cloud_metrics_batch(p1, p2_locals, r, p1_n=None, t1=None, comm=None, partial=False):
    """D1 (and, with normals `p1_n`, D2) metrics between the original cloud p1 (replicated on every rank) and each of several
    candidate decoded clouds, candidate m = union over ranks of `p2_locals[m]`.  Returns one reference-style dictionary per
    candidate (None where the decoded cloud is empty on every rank).  Two collectives for ALL candidates (one MIN, one SUM),
    none in a single process.  Exact for D1 in any world size (squared distances between integer points are integers; A->B is
    a MIN over ranks, B->A a SUM).  D2 across ranks: see sharding.RankGroup.claim."""
    comm = comm or SingleProcess()
    tree_a = t1 if t1 is not None else cKDTree(p1, balanced_tree=False)
    links = []
    for p2 in p2_locals:
        p2 = np.asarray(p2).reshape(-1, 3)
        if len(p2):
            to_b = nearest(cKDTree(p2, balanced_tree=False), p1)
            links.append((p2, to_b, nearest(tree_a, p2), squared_norms(point_gap(p1, p2, to_b))))
        else:
            links.append((p2, np.zeros(len(p1), np.int64), np.zeros(0, np.int64), np.full(len(p1), np.inf)))
    owned = comm.claim([x[3] for x in links], [len(x[0]) > 0 for x in links])
    tallies = np.zeros((len(links), 5), np.float64)
    for m, ((p2, to_b, to_a, _), mine) in enumerate(zip(links, owned)):
        if mine is not None:
            tallies[m] = pair_tally(p1, p2, to_b, to_a, p1_n, None if mine.all() else mine)
    if partial:     # the caller sums the per-rank tallies itself (they ride in a collective it issues anyway) and finishes with finish_metrics
        return tallies, [o is not None for o in owned]
    return finish_metrics(len(p1), comm.total(tallies), [o is not None for o in owned], r, p1_n is not None)


def finish_metrics(n_a, tallies, have, r, with_normals):
    """Reference-style metric dictionaries from the (globally summed) tallies of cloud_metrics_batch(..., partial=True)."""
    groups = GROUPS if with_normals else GROUPS[:1]
    return [metrics_table(n_a, tallies[m], r, groups) if have[m] else None for m in range(len(tallies))]


def compute_metrics(p1, p2, r, p1_n=None, t1=None):
    """The reference's entry point (pc_metric.py:76): metrics of decoded cloud p2 against original p1, peak value r."""
    assert len(p2), 'compute_metrics: empty decoded cloud'
    return cloud_metrics_batch(p1, [p2], r, p1_n, t1)[0]
