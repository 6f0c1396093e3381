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
