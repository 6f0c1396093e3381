"""The CPU oracle (and the product's host-side weight initialisation) at the scaled and wide VTCNN
shapes against the REAL reference's outputs for one training step (tests/golden/shape_*_n2.npz,
tools/make_golden_shapes.py): this pins the oracle where the GPU tests use it as the yardstick
for shapes beyond the headline one."""
import ctypes as C

import numpy as np
import pytest

from conftest import load_golden
from tests_util import compare_compact, shape_case


@pytest.mark.parametrize("name", ["scaled", "wide"])
def test_oracle_step_matches_reference_at_shape(name):
    from oracle.pyoracle import Layout, Oracle, make_dims
    from vit_amd._lib import VtDims, lib
    g = load_golden("shape_%s_n2.npz" % name)
    dims, n, k, seed, x, y = shape_case(g)
    d = make_dims(**dims)
    lo = Layout(d)
    arena = np.zeros(lo.n_total, dtype=np.float32)
    vd = VtDims(dims["image_channels"], dims["input_dim"], dims["tokens"], dims["token_channels"], dims["vt_channels"],
                dims["heads"], dims["num_classes"])
    assert lib.vt_host_init_arena(C.byref(vd), 42, C.c_void_p(arena.ctypes.data)) == 0
    for nm, (off, size, shape, kind) in lo.tensors.items():
        if "p0." + nm in g:
            compare_compact(g, "p0." + nm, arena[off:off + size], 0, exact=True)  # seed-42 draw order, bit-exact
    o = Oracle(d, n)
    p = o.forward(arena, x, training=True)
    compare_compact(g, "fwd.preds", p, 1e-5)
    assert np.array_equal(p.argmax(1), g["fwd.argmax"].ravel())
    gr = o.backward(arena, y)
    # the oracle's parallel loops add in another order than the reference's single thread
    for nm, (off, size, shape, kind) in lo.tensors.items():
        if kind == 0:
            compare_compact(g, "g." + nm, gr[off:off + size], 2e-4)
