"""The record form of the delta behind the projector's BatchNorm (one value + position per row and 16 columns, expanded
on load to coef[N + n] + coef[n] * delta) through the C ABI: vt_gemm_tn_records and vt_gemm_sum_records against numpy
in double precision on the expanded matrix, including the softmax-backward epilogue, several terms, batch strides and
the K-sliced path."""
import ctypes as C

import numpy as np
import pytest

from conftest import rel_err

pytestmark = pytest.mark.gpu


def _records(rng, batch, rows, n):
    """Random records [batch, rows, n / 16, 2] (about a third of the groups empty) and the dense matrix they stand for."""
    ng = n // 16
    val = rng.standard_normal((batch, rows, ng)).astype(np.float32)
    val[rng.random((batch, rows, ng)) < 0.35] = 0.0
    pos = rng.integers(0, 16, size=(batch, rows, ng)).astype(np.int32)
    coef = np.concatenate([rng.uniform(0.5, 1.5, n), 0.1 * rng.standard_normal(n)]).astype(np.float32)
    rec = np.empty((batch, rows, ng, 2), dtype=np.float32)
    rec[..., 0] = val
    rec[..., 1] = pos.view(np.float32)
    delta = np.zeros((batch, rows, n), dtype=np.float64)
    b, r, q = np.meshgrid(np.arange(batch), np.arange(rows), np.arange(ng), indexing="ij")
    delta[b, r, 16 * q + pos] = val
    dense = coef[n:].astype(np.float64) + coef[:n].astype(np.float64) * delta
    return rec, coef, dense


@pytest.mark.parametrize("M,N,K,batch,ws", [(64, 128, 4096, 1, 1 << 22), (16, 128, 1568, 3, 0), (40, 256, 300, 2, 0),
                                             (16, 128, 3001, 2, 1 << 20), (10, 128, 5000, 3, 1 << 20)])  # the 16-row streaming kernel
def test_transposed_product_from_records(M, N, K, batch, ws):
    import torch
    from vit_amd._lib import check, lib
    rng = np.random.default_rng(3)
    rec, coef, dense = _records(rng, batch, K, N)
    A = rng.standard_normal((batch, K, M)).astype(np.float32)
    dA, drec, dcoef = (torch.from_numpy(t).cuda() for t in (A, rec, coef))
    dC = torch.zeros((batch, M, N), dtype=torch.float32, device="cuda")
    wsb = torch.zeros(max(ws, 1), dtype=torch.float32, device="cuda")
    check(lib.vt_gemm_tn_records(M, N, K, C.c_void_p(dA.data_ptr()), M, K * M, C.c_void_p(drec.data_ptr()), K * N // 8,
                                 C.c_void_p(dcoef.data_ptr()), C.c_void_p(dC.data_ptr()), N, M * N, batch,
                                 C.c_void_p(wsb.data_ptr()) if ws else None, ws,
                                 C.c_void_p(torch.cuda.current_stream().cuda_stream)), "vt_gemm_tn_records")
    torch.cuda.synchronize()
    want = np.swapaxes(A.astype(np.float64), 1, 2) @ dense
    assert rel_err(dC.cpu().numpy(), want) < 2e-5


@pytest.mark.parametrize("K0", [128, 96, 80])  # eight and six groups applied row by row; five expanded into the matrix steps
def test_sum_and_softmax_backward_from_records(K0):
    import torch
    from vit_amd._lib import VtGemmTerm, check, lib
    rng = np.random.default_rng(4)
    M, batch = 4096, 2
    rec, coef, dense = _records(rng, batch, M, K0)
    drec, dcoef = torch.from_numpy(rec).cuda(), torch.from_numpy(coef).cuda()
    s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    # softmax backward of (dz . values^T) against an attention map, 16 columns
    vals = rng.standard_normal((batch, 16, K0)).astype(np.float32)
    att = rng.random((batch, M, 16)).astype(np.float32)
    att /= att.sum(axis=2, keepdims=True)
    dvals, datt = torch.from_numpy(vals).cuda(), torch.from_numpy(att).cuda()
    out = torch.zeros((batch, M, 16), dtype=torch.float32, device="cuda")
    t = (VtGemmTerm * 1)()
    t[0] = VtGemmTerm(drec.data_ptr(), dvals.data_ptr(), M * K0 // 8, 16 * K0, K0, K0 // 8, K0, 1)
    check(lib.vt_gemm_sum_records(1, t, C.c_void_p(dcoef.data_ptr()), 2, C.c_void_p(datt.data_ptr()), M, 16,
                                  C.c_void_p(out.data_ptr()), 16, M * 16, None, batch, s), "vt_gemm_sum_records")
    torch.cuda.synchronize()
    prod = dense @ np.swapaxes(vals.astype(np.float64), 1, 2)
    a64 = att.astype(np.float64)
    want = a64 * (prod - (a64 * prod).sum(axis=2, keepdims=True))
    assert rel_err(out.cpu().numpy(), want) < 2e-5
    # the same with 196 columns (thirteen result tiles, the values' copy in LDS beyond 64 KB)
    vals2 = rng.standard_normal((batch, 196, K0)).astype(np.float32)
    att2 = rng.random((batch, M, 196)).astype(np.float32)
    att2 /= att2.sum(axis=2, keepdims=True)
    dvals2, datt2 = torch.from_numpy(vals2).cuda(), torch.from_numpy(att2).cuda()
    out2 = torch.zeros((batch, M, 196), dtype=torch.float32, device="cuda")
    t[0] = VtGemmTerm(drec.data_ptr(), dvals2.data_ptr(), M * K0 // 8, 196 * K0, K0, K0 // 8, K0, 1)
    check(lib.vt_gemm_sum_records(1, t, C.c_void_p(dcoef.data_ptr()), 2, C.c_void_p(datt2.data_ptr()), M, 196,
                                  C.c_void_p(out2.data_ptr()), 196, M * 196, None, batch, s), "vt_gemm_sum_records")
    torch.cuda.synchronize()
    prod2 = dense @ np.swapaxes(vals2.astype(np.float64), 1, 2)
    b64 = att2.astype(np.float64)
    assert rel_err(out2.cpu().numpy(), b64 * (prod2 - (b64 * prod2).sum(axis=2, keepdims=True))) < 2e-5
    t[0] = VtGemmTerm(drec.data_ptr(), dvals.data_ptr(), M * K0 // 8, 16 * K0, K0, K0 // 8, K0, 1)
    # a sum of three products into 64 columns, the first from records, with a bias
    W0 = rng.standard_normal((64, K0)).astype(np.float32)          # shared, read transposed
    A1 = rng.standard_normal((batch, M, 16)).astype(np.float32)
    B1 = rng.standard_normal((batch, 16, 64)).astype(np.float32)    # per batch member
    A2 = rng.standard_normal((batch, M, 16)).astype(np.float32)
    B2 = rng.standard_normal((64, 16)).astype(np.float32)           # shared, read transposed
    bias = rng.standard_normal(64).astype(np.float32)
    d = [torch.from_numpy(x).cuda() for x in (W0, A1, B1, A2, B2, bias)]
    t3 = (VtGemmTerm * 3)()
    t3[0] = VtGemmTerm(drec.data_ptr(), d[0].data_ptr(), M * K0 // 8, 0, K0, K0 // 8, K0, 1)
    t3[1] = VtGemmTerm(d[1].data_ptr(), d[2].data_ptr(), M * 16, 16 * 64, 16, 16, 64, 0)
    t3[2] = VtGemmTerm(d[3].data_ptr(), d[4].data_ptr(), M * 16, 0, 16, 16, 16, 1)
    out3 = torch.zeros((batch, M, 64), dtype=torch.float32, device="cuda")
    check(lib.vt_gemm_sum_records(3, t3, C.c_void_p(dcoef.data_ptr()), 0, None, M, 64, C.c_void_p(out3.data_ptr()), 64, M * 64,
                                  C.c_void_p(d[5].data_ptr()), batch, s), "vt_gemm_sum_records")
    torch.cuda.synchronize()
    want3 = (dense @ W0.astype(np.float64).T + A1.astype(np.float64) @ B1.astype(np.float64)
             + A2.astype(np.float64) @ B2.astype(np.float64).T + bias)
    assert rel_err(out3.cpu().numpy(), want3) < 2e-5
    # shapes the record kernels do not take are refused, nothing launched
    from vit_amd._lib import lib as L
    assert L.vt_gemm_sum_records(1, t, C.c_void_p(dcoef.data_ptr()), 0, None, 64, 16, C.c_void_p(out.data_ptr()), 16, 64 * 16, None,
                                 1, s) == -2  # fewer than 2048 rows


def test_batchnorm_apply_inside_its_first_product():
    """vt_gemm_bn_apply: the normalised matrix and the product that reads it from one pass over the BatchNorm input."""
    import torch
    from vit_amd._lib import check, lib
    rng = np.random.default_rng(6)
    M, N, K = 5000, 16, 64
    A = (rng.standard_normal((M, K)) * 3 + 1).astype(np.float32)
    stat = np.concatenate([A.mean(0), 1.0 / np.sqrt(A.var(0) + 1e-5)]).astype(np.float32)
    gamma, beta = rng.uniform(0.5, 1.5, K).astype(np.float32), rng.standard_normal(K).astype(np.float32)
    B, bias = rng.standard_normal((K, N)).astype(np.float32), rng.standard_normal(N).astype(np.float32)
    d = [torch.from_numpy(x).cuda() for x in (A, stat, gamma, beta, B, bias)]
    x = torch.zeros((M, K), dtype=torch.float32, device="cuda")
    out = torch.zeros((M, N), dtype=torch.float32, device="cuda")
    P = lambda t: C.c_void_p(t.data_ptr())
    check(lib.vt_gemm_bn_apply(M, N, K, P(d[0]), K, P(d[1]), P(d[2]), P(d[3]), P(x), P(d[4]), N, 0, P(d[5]), P(out), N,
                               C.c_void_p(torch.cuda.current_stream().cuda_stream)), "vt_gemm_bn_apply")
    torch.cuda.synchronize()
    want_x = (A.astype(np.float64) - stat[:K]) * stat[K:] * gamma + beta
    assert rel_err(x.cpu().numpy(), want_x) < 2e-6
    assert rel_err(out.cpu().numpy(), want_x @ B.astype(np.float64) + bias) < 2e-6
    # the same launch leaving per-tile records of the softmax over rows (maximum, sum of exp relative to it)
    tiles = (M + 15) // 16
    for Nn, dB_, dbias_, Bn, biasn in ((N, d[4], d[5], B, bias),):
        rec = torch.full((tiles, 2, Nn), -7.0, dtype=torch.float32, device="cuda")
        x3 = torch.zeros((M, K), dtype=torch.float32, device="cuda")
        out3 = torch.zeros((M, Nn), dtype=torch.float32, device="cuda")
        check(lib.vt_gemm_bn_apply_tile_records(M, Nn, K, P(d[0]), K, P(d[1]), P(d[2]), P(d[3]), P(x3), P(dB_), Nn, 0, P(dbias_),
                                                P(out3), Nn, P(rec), C.c_void_p(torch.cuda.current_stream().cuda_stream)),
              "vt_gemm_bn_apply_tile_records")
        torch.cuda.synchronize()
        assert torch.equal(out3, out) and torch.equal(x3, x)
        S = out3.cpu().numpy().astype(np.float64)
        pad = np.full((tiles * 16 - M, Nn), -np.inf)
        St = np.concatenate([S, pad]).reshape(tiles, 16, Nn)
        tm = St.max(axis=1)
        tz = np.exp(St - tm[:, None, :]).sum(axis=1)
        r = rec.cpu().numpy()
        assert np.array_equal(r[:, 0, :], tm.astype(np.float32))
        assert rel_err(r[:, 1, :], tz) < 2e-6
    # 17 ... 128 result columns: refused, nothing launched
    assert lib.vt_gemm_bn_apply(M, 32, K, P(d[0]), K, P(d[1]), P(d[2]), P(d[3]), P(x), P(d[4]), 32, 0, None, P(out), 32, None) == -2
    # 196 result columns (thirteen tiles)
    B2, bias2 = rng.standard_normal((K, 196)).astype(np.float32), rng.standard_normal(196).astype(np.float32)
    dB2, dbias2 = torch.from_numpy(B2).cuda(), torch.from_numpy(bias2).cuda()
    x2 = torch.zeros((M, K), dtype=torch.float32, device="cuda")
    out2 = torch.zeros((M, 196), dtype=torch.float32, device="cuda")
    check(lib.vt_gemm_bn_apply(M, 196, K, P(d[0]), K, P(d[1]), P(d[2]), P(d[3]), P(x2), P(dB2), 196, 0, P(dbias2), P(out2), 196,
                               C.c_void_p(torch.cuda.current_stream().cuda_stream)), "vt_gemm_bn_apply")
    torch.cuda.synchronize()
    assert rel_err(x2.cpu().numpy(), want_x) < 2e-6
    assert rel_err(out2.cpu().numpy(), want_x @ B2.astype(np.float64) + bias2) < 2e-6
    rec2 = torch.full((tiles, 2, 196), -7.0, dtype=torch.float32, device="cuda")
    out4 = torch.zeros((M, 196), dtype=torch.float32, device="cuda")
    check(lib.vt_gemm_bn_apply_tile_records(M, 196, K, P(d[0]), K, P(d[1]), P(d[2]), P(d[3]), P(x2), P(dB2), 196, 0, P(dbias2),
                                            P(out4), 196, P(rec2), C.c_void_p(torch.cuda.current_stream().cuda_stream)),
          "vt_gemm_bn_apply_tile_records")
    torch.cuda.synchronize()
    assert torch.equal(out4, out2)
    S = out4.cpu().numpy().astype(np.float64)
    St = np.concatenate([S, np.full((tiles * 16 - M, 196), -np.inf)]).reshape(tiles, 16, 196)
    tm = St.max(axis=1)
    r = rec2.cpu().numpy()
    assert np.array_equal(r[:, 0, :], tm.astype(np.float32))
    assert rel_err(r[:, 1, :], np.exp(St - tm[:, None, :]).sum(axis=1)) < 2e-6
