"""Stage-by-stage parity of the GPU compress graph against the CPU oracle, with NO conditional asserts.

The graph alternates float stages (conv stacks) and integer decisions (rounding, scale -> table index).  Two correct fp32
implementations differ by round-off in the float stages, so a free-running comparison can legitimately flip an integer
decision that sits on a boundary -- and everything downstream of it.  Every stage is therefore checked on the GPU's OWN
upstream tensors ("teacher forcing"): float stages within the stated tolerance, integer stages and bytes bit-exact.  The
free-running oracle is compared too, and each of its integer disagreements must be a proven boundary case.

Reference wiring: /root/reference/src/model_types.py:283-309 (V1), :371-411 (V2).
"""
import numpy as np

STACK_TOL = 1e-4        # whole conv stack: |gpu - oracle| <= STACK_TOL * (1 + max|oracle|); the reference's own enc/dec
                        # tolerance is 1e-3 (decompress_octree.py:94)


def oracle_model(model, name):
    eb, m = model.entropy_bottleneck, dict(config=name, params=model.get_weights(), round_mode=model.round_mode,
                                           data_format=model.data_format)
    m['eb'] = dict(cdf=eb.quantized_cdf, cdf_size=eb.cdf_length, offset=eb.offset, medians=eb.medians)
    if getattr(model, 'conditional_bottleneck', None) is not None:
        gc = model.conditional_bottleneck
        m['gc'] = (gc.quantized_cdf, gc.cdf_length, gc.offset)
        m['scale_table'] = gc.scale_table_f32
    return m


def _close(got, ref, what, tol=STACK_TOL):
    err = float(np.abs(np.asarray(got, np.float64) - np.asarray(ref, np.float64)).max())
    bound = tol * (1 + float(np.abs(ref).max()))
    assert err <= bound, f'{what}: max abs error {err:.3e} > {bound:.3e}'
    return bound


def check_block(O, om, dense, g, strings, run=None, tol=STACK_TOL):
    """dense: (D,H,W) float32 occupancy of ONE block; g: the GPU's debug dict of that block (model._encode_batch(debug=True));
    strings: the GPU's strings of that block.  `run`: conv-stack backend of the oracle (None = the C loops,
    oracle.torch_oracle.run_transform for 64^3 blocks, run_transform_fp16 for the fp16 mode); `tol`: tolerance of the float stages.
    Returns a dict of diagnostic counts."""
    run = run or O.run_transform
    cfg = O.CONFIGS[om['config']]
    P, F, rm = om['params'], cfg['F'], om.get('round_mode', 0)
    eb = om['eb']
    x = dense[None, ..., None].astype(np.float32)
    dhw = dense.shape
    info = {}

    # ---- analysis transform (float)
    y_o = np.asarray(run(cfg['a'], F, P, 'analysis', x), np.float32)
    tol_y = _close(g['y'], y_o, 'analysis output y', tol)

    if cfg['v'] == 1:
        # ---- quantise y (integer, on the GPU's own y): bit-exact; bytes bit-exact
        sym_o, yhat_o = O.quantize(g['y'], eb['medians'], rm)
        assert np.array_equal(sym_o, g['symbols']), 'EntropyBottleneck symbols differ on identical input'
        assert np.array_equal(yhat_o, g['y_hat']), 'dequantised y_hat differs on identical input'
        y_string = O.range_encode(O.to_stream(g['symbols'], om), O._channel_rows(g['y'].shape, om), eb['cdf'], eb['cdf_size'],
                                  eb['offset'])
        assert strings[0] == y_string, 'y_string bytes differ from the oracle coder on identical symbols'
        # free-running oracle symbols: every disagreement must sit on a rounding boundary
        sym_f, _ = O.quantize(y_o, eb['medians'], rm)
        info['sym_flips'] = _boundary_only_sym(sym_f, g['symbols'], y_o, eb['medians'], tol_y, rm)
    else:
        # ---- hyper-analysis (float, on the GPU's y)
        z_o = np.asarray(run('HyperAnalysisTransform', F, P, 'hyper_analysis', g['y']), np.float32)
        _close(g['z'], z_o, 'hyper-analysis output z', tol)
        # ---- quantise z (integer, on the GPU's z), z_string bytes
        zsym_o, zhat_o = O.quantize(g['z'], eb['medians'], rm)
        assert np.array_equal(zsym_o, g['z_symbols']), 'z symbols differ on identical input'
        assert np.array_equal(zhat_o, g['z_hat']), 'z_hat differs on identical input'
        z_string = O.range_encode(O.to_stream(g['z_symbols'], om), O._channel_rows(g['z'].shape, om), eb['cdf'], eb['cdf_size'],
                                  eb['offset'])
        assert strings[1] == z_string, 'z_string bytes differ from the oracle coder on identical symbols'
        # ---- hyper-synthesis (float, on the GPU's z_hat), scale -> index (integer, on the GPU's sigma)
        sig_o = np.asarray(run('HyperSynthesisTransform', F, P, 'hyper_synthesis', g['z_hat']), np.float32)
        tol_s = _close(g['sigma_hat'], sig_o, 'hyper-synthesis output sigma_hat', tol)
        idx_o = O.scale_index(g['sigma_hat'], om['scale_table'])
        assert np.array_equal(idx_o, g['indexes']), 'scale indexes differ on identical sigma_hat'
        # free-running oracle indexes: disagreements only where the oracle's sigma sits within the float tolerance of a table entry
        idx_f = O.scale_index(sig_o, om['scale_table'])
        bad = np.flatnonzero(idx_f.ravel() != g['indexes'].ravel())
        if len(bad):
            # the GPU's index j (first j with sigma <= table[j]) must be the index of SOME sigma within the float tolerance of the
            # oracle's: table[j - 1] - tol < sigma_oracle <= table[j] + tol (at fp32 tolerances that is "off by exactly one, next
            # to a table entry"; the fp16 mode's tolerance can span two of the closely spaced low entries)
            tab = np.asarray(om['scale_table'], np.float64)
            sv = np.maximum(sig_o.ravel()[bad].astype(np.float64), tab[0])
            j = g['indexes'].ravel()[bad].astype(np.int64)
            lo = np.where(j > 0, tab[np.maximum(j - 1, 0)], -np.inf)
            hi = np.where(j < len(tab) - 1, tab[j], np.inf)
            assert np.all((lo - tol_s < sv) & (sv <= hi + tol_s)), 'scale index differs away from a table boundary'
            if tol <= STACK_TOL:
                assert np.all(np.abs(idx_f.ravel()[bad] - j) == 1)
        info['idx_flips'] = int(len(bad))
        # ---- quantise y (integer, on the GPU's y), y_string bytes with the GPU's indexes
        ysym_o, yhat_o = O.quantize(g['y'], None, rm)
        assert np.array_equal(ysym_o, g['symbols']), 'GaussianConditional symbols differ on identical input'
        assert np.array_equal(yhat_o, g['y_hat']), 'y_hat differs on identical input'
        gcdf, gsize, goff = om['gc']
        y_string = O.range_encode(O.to_stream(g['symbols'], om), O.to_stream(g['indexes'], om), gcdf, gsize, goff)
        assert strings[0] == y_string, 'y_string bytes differ from the oracle coder on identical symbols and indexes'
        ysym_f, _ = O.quantize(y_o, None, rm)
        info['sym_flips'] = _boundary_only_sym(ysym_f, g['symbols'], y_o, None, tol_y, rm)

    # ---- synthesis (float, on the GPU's y_hat)
    xhat_o = np.asarray(run(cfg['s'], F, P, 'synthesis', g['y_hat']), np.float32)
    _close(g['x_hat'], xhat_o, 'synthesis output x_hat', tol)

    # ---- the ORACLE DECODER parses the GPU's strings (the interoperability direction): z exactly; the oracle's own sigma /
    #      indexes may differ from the encoder's only on proven boundaries (checked above on the same z_hat), so y is decoded
    #      with the encoder's indexes, must reproduce the encoder's symbols exactly, and the oracle's synthesis of them must
    #      match the GPU's x_hat within the tolerance
    if cfg['v'] == 1:
        xd, dd = O.decompress_block(om, strings, dhw, run=run)
        assert np.array_equal(dd['symbols'], g['symbols']) and np.array_equal(dd['y_hat'], g['y_hat'])
    else:
        xd, dd = O.decompress_block(om, strings, dhw, run=run, indexes=g['indexes'])
        assert np.array_equal(dd['z_symbols'], g['z_symbols']) and np.array_equal(dd['z_hat'], g['z_hat'])
        assert np.array_equal(dd['symbols'], g['symbols']) and np.array_equal(dd['y_hat'], g['y_hat'])
        own_bad = int(np.count_nonzero(dd['own_indexes'] != g['indexes']))
        assert own_bad == info['idx_flips'], 'decoder-side oracle indexes disagree with the encoder-side oracle run'
    _close(g['x_hat'][0, ..., 0], xd, 'oracle-decoded x_hat', tol)
    info['x_hat_max'] = float(np.abs(xhat_o).max())
    return info


def _boundary_only_sym(sym_free, sym_gpu, v_free, medians, tol, mode):
    """symbols of the free-running oracle vs the GPU's: they may differ (by one) only where the oracle's float input is
    within `tol` of a rounding boundary (mode 0: floor(v + 0.5 - median); mode 1: round-half-even(v - median): both break at
    v - median = k + 0.5)."""
    bad = np.flatnonzero(sym_free.ravel() != sym_gpu.ravel())
    if len(bad):
        v = v_free.astype(np.float64)
        if medians is not None:
            v = v - np.asarray(medians, np.float64)
        t = v.ravel()[bad] - 0.5
        assert np.all(np.abs(t - np.round(t)) <= tol), 'symbol mismatch away from a rounding boundary'
        assert np.all(np.abs(sym_free.ravel()[bad] - sym_gpu.ravel()[bad]) == 1)
    return int(len(bad))
