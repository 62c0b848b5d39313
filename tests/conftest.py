import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


@pytest.fixture(scope='session')
def oracle():
    from oracle import oracle as O
    O.lib()
    return O


@pytest.fixture(scope='session')
def ctx():
    import torch
    if not torch.cuda.is_available():
        pytest.skip('no GPU visible')
    from pcc_geo_cnn_v2_amd import ops
    return ops.Context(0)


@pytest.fixture(autouse=True)
def _numerics_back_to_default(request):
    """Tests switch kernel families through ctx.set_numerics / ctx.numerics_override (include/pcc_geo.h, "codec numerics"): whatever a
    test leaves behind -- also when it fails half-way -- is undone, on the session context and on the package's cached ones."""
    yield
    if 'ctx' not in request.fixturenames:
        return
    import ctypes as C
    from pcc_geo_cnn_v2_amd import _lib as L, ops
    for c in [request.getfixturevalue('ctx')] + list(ops._CONTEXTS.values()):
        L.lib().pcc_ctx_set_numerics(c.handle, C.c_uint32(c.numerics_at_creation))
