"""The numpy restatement of SelfAttention (oracle/selfattn.py) against fixtures from the real reference
layer: pins the checker that the GPU test's tolerances are judged by."""
import os

import numpy as np
import pytest

from conftest import rel_err

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.mark.parametrize("case", ["small", "tokens", "ragged"])
def test_selfattn_oracle_matches_reference_fixture(case):
    from oracle import selfattn
    g = np.load(os.path.join(GOLD, "selfattn_%s.npz" % case))
    out, _, _, attn = selfattn.forward(g["x"], g["key.W"], g["query.W"])
    assert np.allclose(attn.sum(-1), 1.0, atol=1e-5)
    assert rel_err(out, g["out"]) < 2e-6
    g_wk, g_wq, dx = selfattn.backward(g["x"], g["key.W"], g["query.W"], g["delta"])
    assert rel_err(g_wk, g["g.key.W"]) < 5e-6
    assert rel_err(g_wq, g["g.query.W"]) < 5e-6
    assert rel_err(dx, g["dx"]) < 5e-6
