"""Entropy models of the codec: the factorized prior (tfc.EntropyBottleneck) and the scale
hyperprior (tfc.GaussianConditional as patched by the reference).

Table construction follows /root/reference/src/utils/patch_gaussian_conditional.py:49-125 for the
Gaussian conditional and the published tensorflow-compression 1.3 `EntropyBottleneck.build` for the
factorized prior (source not under /root/reference: requirements.txt:7).  Tables are built on the host
once per model (the reference stores them as non-trainable checkpoint variables, patch...:91-97) and
can be loaded verbatim from a checkpoint instead.
"""
import numpy as np
from scipy.special import erfc
from scipy.stats import norm

from . import ops


def scale_table(scales_min=0.11, scales_max=256, scales_levels=64):
    """src/model_types.py:324"""
    return np.exp(np.linspace(np.log(scales_min), np.log(scales_max), scales_levels))


class GaussianConditional:
    """Zero-mean Gaussian with per-element scale picked from a fixed log-spaced table.

    tail_mass: tfc 1.3's `EntropyModel.__init__` default is 2**-8 (the reference never overrides it,
    src/model_types.py:385,406); SURVEY.md quotes 1e-9, which is tfc's `likelihood_bound`.  It only
    changes the width of the stored tables and is a constructor argument here.
    """

    def __init__(self, table, tail_mass=2 ** -8, range_coder_precision=16, tables=None):
        self.scale_table = np.asarray(table, np.float64)
        self.scale_table_f32 = self.scale_table.astype(np.float32)
        self.tail_mass = tail_mass
        self.precision = range_coder_precision
        if tables is None:
            tables = self._build()
        self.quantized_cdf, self.cdf_length, self.offset = tables
        self.table = ops.HostCdfTable(self.quantized_cdf, self.cdf_length, self.offset, self.precision, 4)

    def _build(self):
        multiplier = -norm.ppf(self.tail_mass / 2)                                  # patch...:62
        pmf_center = np.ceil(self.scale_table * multiplier).astype(int)             # :63
        pmf_length = 2 * pmf_center + 1
        max_length = int(np.max(pmf_length))
        samples = np.abs(np.arange(max_length, dtype=int) - pmf_center[:, None]).astype(np.float32)   # :73
        sc = self.scale_table_f32[:, None]
        c = np.float32(-(2 ** -0.5))
        upper = (np.float32(.5) * erfc(c * ((np.float32(.5) - samples) / sc))).astype(np.float32)     # :76
        lower = (np.float32(.5) * erfc(c * ((np.float32(-.5) - samples) / sc))).astype(np.float32)    # :77
        pmf = (upper - lower).astype(np.float32)
        tail = (2 * lower[:, :1]).astype(np.float32)                                # :81
        cdf = np.zeros((len(self.scale_table), max_length + 2), np.int32)
        for i in range(len(self.scale_table)):                                      # _pmf_to_cdf, :87-89
            prob = np.concatenate([pmf[i, :pmf_length[i]], tail[i]]).astype(np.float32)
            cdf[i, :pmf_length[i] + 2] = ops.pmf_to_quantized_cdf(prob, self.precision)
        return cdf, (pmf_length + 2).astype(np.int32), (-pmf_center).astype(np.int32)   # :96,:118


def _softplus(x):
    return np.logaddexp(0, x)


class EntropyBottleneck:
    """Factorized prior: per-channel non-parametric density (Balle et al. 2018), tfc 1.3 layout:
    matrices[i] (C, f[i+1], f[i]), biases[i] (C, f[i+1], 1), factors[i] (C, f[i+1], 1) with
    f = (1,) + filters + (1,), quantiles (C, 1, 3)."""

    def __init__(self, channels, init_scale=10, filters=(3, 3, 3), range_coder_precision=16, params=None,
                 tables=None, seed=42):
        self.channels = int(channels)
        self.filters = tuple(filters)
        self.precision = range_coder_precision
        self.params = params if params is not None else self.init_params(channels, init_scale, self.filters, seed)
        if tables is None:
            tables = self._build()
        self.quantized_cdf, self.cdf_length, self.offset = tables
        self.medians = np.ascontiguousarray(self.params['quantiles'][:, 0, 1], np.float32)
        self.table = ops.HostCdfTable(self.quantized_cdf, self.cdf_length, self.offset, self.precision, 4)

    @staticmethod
    def init_params(channels, init_scale=10, filters=(3, 3, 3), seed=42):
        """tfc 1.3 EntropyBottleneck.build initialisers (biases: uniform(-.5,.5), seeded)."""
        rng = np.random.default_rng(seed)
        f = (1,) + tuple(filters) + (1,)
        scale = init_scale ** (1 / (len(filters) + 1))
        p = {}
        for i in range(len(filters) + 1):
            init = np.log(np.expm1(1 / scale / f[i + 1]))
            p[f'matrix_{i}'] = np.full((channels, f[i + 1], f[i]), init, np.float32)
            p[f'bias_{i}'] = rng.uniform(-.5, .5, (channels, f[i + 1], 1)).astype(np.float32)
            if i < len(filters):
                p[f'factor_{i}'] = np.zeros((channels, f[i + 1], 1), np.float32)
        p['quantiles'] = np.tile(np.array([[[-init_scale, 0, init_scale]]], np.float32), (channels, 1, 1))
        return p

    def _logits_cumulative(self, x):
        logits = x.astype(np.float32)
        n = len(self.filters) + 1
        for i in range(n):
            m = _softplus(self.params[f'matrix_{i}'].astype(np.float32)).astype(np.float32)
            logits = (np.matmul(m, logits) + self.params[f'bias_{i}']).astype(np.float32)
            if i < n - 1:
                logits = (logits + np.tanh(self.params[f'factor_{i}']) * np.tanh(logits)).astype(np.float32)
        return logits

    def _build(self):
        q = self.params['quantiles'].astype(np.float32)
        medians = q[:, 0, 1]
        minima = np.maximum(np.ceil(medians - q[:, 0, 0]).astype(np.int32), 0)
        maxima = np.maximum(np.ceil(q[:, 0, 2] - medians).astype(np.int32), 0)
        pmf_start = medians - minima.astype(np.float32)
        pmf_length = maxima + minima + 1
        max_length = int(pmf_length.max())
        samples = np.arange(max_length, dtype=np.float32)[None, None, :] + pmf_start[:, None, None]
        half = np.float32(.5)
        lower = self._logits_cumulative(samples - half)
        upper = self._logits_cumulative(samples + half)
        sig = lambda t: (1.0 / (1.0 + np.exp(-t.astype(np.float64)))).astype(np.float32)
        sign = -np.sign(lower + upper)
        pmf = np.abs(sig(sign * upper) - sig(sign * lower))[:, 0, :]
        tail = (sig(lower[:, 0, :1]) + sig(-upper[:, 0, -1:])).astype(np.float32)
        cdf = np.zeros((self.channels, max_length + 2), np.int32)
        for c in range(self.channels):
            prob = np.concatenate([pmf[c, :pmf_length[c]], tail[c]]).astype(np.float32)
            cdf[c, :pmf_length[c] + 2] = ops.pmf_to_quantized_cdf(prob, self.precision)
        return cdf, (pmf_length + 2).astype(np.int32), (-minima).astype(np.int32)
