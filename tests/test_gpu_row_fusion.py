"""The projector's token-side backward and the transformer layer's backward step a as ONE launch
(vt_projector_tokens_bwd_transformer_a) against the two separate launchers, through the C ABI: every
output must be bit-identical (same arithmetic, same order), on full and ragged row counts."""
import ctypes as C

import pytest

D, C1, C2, PT_STRIDE, RB = 32, 8, 16, 32 * 32 + 32 * 16 + 16, 64


@pytest.mark.gpu
@pytest.mark.parametrize("rows", [16 * 1024, 16 * 5, 16 * 3 + 7, 1])
def test_fused_token_backward_is_bit_identical(rows):
    import torch
    from vit_amd._lib import check, lib
    gen = torch.Generator(device="cuda").manual_seed(rows)
    rnd = lambda *s: torch.randn(*s, device="cuda", generator=gen)  # noqa: E731
    o, dv, dk, s2 = rnd(rows, D), rnd(rows, C2), rnd(rows, D), rnd(rows, D)
    wv, wk, w2 = rnd(D * C2), rnd(D * D), rnd(C1 * D)
    stat = torch.cat([rnd(D), rnd(D).abs() + 0.5])
    nb = (rows + RB - 1) // RB
    pv = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)

    def outputs():
        return [torch.full((rows, D), 3.0, device="cuda"), torch.full((PT_STRIDE,), 3.0, device="cuda"),
                torch.full((nb * 2 * D,), 3.0, device="cuda"), torch.full((rows, C1), 3.0, device="cuda"),
                torch.empty(nb * PT_STRIDE, device="cuda")]

    pdt, g, part, dT0, scratch = a = outputs()
    check(lib.vt_projector_tokens_bwd(pv(o), pv(dv), pv(dk), pv(wv), pv(wk), pv(pdt), pv(g), pv(scratch), rows, st))
    check(lib.vt_transformer_bwd_a(pv(o), pv(pdt), pv(s2), pv(stat), pv(w2), pv(part), pv(dT0), rows, st))
    pdt, g, part, dT0, scratch = b = outputs()
    check(lib.vt_projector_tokens_bwd_transformer_a(pv(o), pv(dv), pv(dk), pv(wv), pv(wk), pv(pdt), pv(g), pv(scratch), pv(s2),
                                                    pv(stat), pv(w2), pv(part), pv(dT0), rows, st))
    torch.cuda.synchronize()
    for name, x, y in zip(("token delta", "gradient block", "norm2 partial sums", "tokenizer delta"), a, b):
        assert torch.equal(x, y), name
        assert torch.isfinite(x).all() and float(x.abs().max()) > 0, name
    assert lib.vt_projector_tokens_bwd_transformer_a(pv(o), pv(dv), pv(dk), pv(wv), pv(wk), pv(pdt), pv(g), pv(scratch), pv(s2),
                                                     pv(stat), pv(w2), pv(part), pv(dT0), 0, st) != 0


@pytest.mark.gpu
@pytest.mark.parametrize("n,k", [(1, 8), (5, 10), (32, 8), (200, 3), (1024, 8)])
def test_pair_form_head_backward_equals_the_dense_one(n, k):
    """vt_head_bwd_windows (delta per element pair, ReLU gate and x_hat from the window maxima the forward kept,
    window-major workgroups) against vt_head_bwd (dense delta, re-reading z), both fed by vt_head_fwd's outputs:
    the expanded delta equals the dense one up to rounding of the K-term sum (its lowest mantissa bit carries the arg-max column),
    d_logits are identical, the BatchNorm partial sums agree to rounding; and vt_projector_bwd_windows equals
    vt_projector_bwd on the expanded delta."""
    import numpy as np
    import torch
    from vit_amd._lib import check, lib
    P, Q, H = 784, C2 * 196, 28
    gen = torch.Generator(device="cuda").manual_seed(1000 * k + n)
    rnd = lambda *s: torch.randn(*s, device="cuda", generator=gen)  # noqa: E731
    pv = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    z = rnd(n, P, C2)
    stat = torch.cat([0.1 * rnd(C2), rnd(C2).abs() + 0.5])
    gamma, beta = rnd(C2).abs() + 0.3, 0.2 * rnd(C2)
    wfc, bfc = 0.05 * rnd(Q, k), 0.1 * rnd(k)
    labels = torch.randint(0, k, (n,), device="cuda", generator=gen, dtype=torch.int32)
    pooled = torch.empty(n, Q, device="cuda")
    pix = torch.empty(n * Q, device="cuda", dtype=torch.uint8)
    logits, probs = torch.empty(n, k, device="cuda"), torch.empty(n, k, device="cuda")
    check(lib.vt_head_fwd(pv(z), pv(stat), pv(gamma), pv(beta), pv(wfc), pv(bfc), pv(pooled), pv(pix), pv(logits), pv(probs),
                          n, H, k, st))
    dl_a, du_a, part_a = torch.empty(n, k, device="cuda"), torch.empty(n, P, C2, device="cuda"), torch.empty(n, 2 * C2, device="cuda")
    check(lib.vt_head_bwd(pv(probs), pv(labels), pv(wfc), pv(pix), pv(z), pv(stat), pv(gamma), pv(beta), pv(dl_a), pv(du_a),
                          pv(part_a), n, H, k, st))
    blocks = lib.vt_head_bwd_windows_blocks(n)
    assert blocks >= 13
    dl_b, dpair = torch.empty(n, k, device="cuda"), torch.empty(n, P * C2 // 2, device="cuda")
    part_b, du_b = torch.empty(blocks, 2 * C2, device="cuda"), torch.empty(n, P, C2, device="cuda")
    check(lib.vt_head_bwd_windows(pv(probs), pv(labels), pv(wfc), pv(pix), pv(pooled), pv(z), pv(stat), pv(gamma), pv(beta),
                                  pv(dl_b), pv(dpair), pv(part_b), n, H, k, st))
    check(lib.vt_head_expand_windows(pv(dpair), pv(du_b), n, H, st))
    torch.cuda.synchronize()
    assert torch.equal(dl_a, dl_b)
    a_np, b_np = du_a.cpu().numpy(), du_b.cpu().numpy()
    nz = a_np != 0
    assert nz.sum() > 0 and np.array_equal(b_np != 0, nz)  # the same elements receive a delta
    # the same value up to the order of the K-term sum and the lowest mantissa bit (cleared in the pair form)
    assert np.abs(a_np[nz] - b_np[nz]).max() <= 4e-6 * np.abs(a_np[nz]).max()
    assert not (b_np.view(np.uint32) & np.uint32(1)).any()
    sa = part_a.double().sum(0).cpu().numpy()
    sb = part_b.double().sum(0).cpu().numpy()
    assert np.abs(sa - sb).max() <= 2e-5 * np.abs(sa).max()
    # the projector backward fed by either form of the same delta
    L, D = 16, 32
    conv, coef = rnd(n, P, C1), torch.cat([rnd(C2).abs() + 0.5, 0.01 * rnd(C2)])
    bn_stat, bn_g, bn_b = torch.cat([0.1 * rnd(C1), rnd(C1).abs() + 0.5]), rnd(C1).abs() + 0.3, 0.2 * rnd(C1)
    pk = 0.3 * rnd(n, C1 * L + L * C2 + L * D)
    wq, wd = 0.3 * rnd(C1, D), 0.3 * rnd(C1, C2)

    def run(windows):
        out = [torch.empty(n, P, C1, device="cuda"), torch.empty(n, L, C2, device="cuda"), torch.empty(n, L, D, device="cuda"),
               torch.empty(n, 400, device="cuda")]
        if windows:
            check(lib.vt_projector_bwd_windows(pv(dpair), pv(coef), pv(conv), pv(bn_stat), pv(bn_g), pv(bn_b), pv(pk), pv(wq), pv(wd),
                                               *[pv(t) for t in out], n, P, st))
        else:
            check(lib.vt_projector_bwd(pv(du_b), pv(coef), pv(conv), pv(bn_stat), pv(bn_g), pv(bn_b), pv(pk), pv(wq), pv(wd),
                                       *[pv(t) for t in out], n, P, st))
        torch.cuda.synchronize()
        return out
    for name, x, y in zip(("input delta", "value delta", "key delta", "gradient slab"), run(False), run(True)):
        assert torch.equal(x, y), name
        assert torch.isfinite(x).all()
