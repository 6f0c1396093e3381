"""vt_gemm_f32 through the C ABI against numpy in double precision: every transpose combination,
ragged tile edges, leading dimensions wider than the matrix, batch strides, bias, accumulation and
the K-sliced path (long contraction, small result) that the weight gradients use."""
import ctypes as C

import numpy as np
import pytest

from conftest import rel_err

pytestmark = pytest.mark.gpu


def _run(ta, tb, M, N, K, batch=1, alpha=1.0, bias=False, acc=False, pad=0, ws_floats=0, seed=0, tol=2e-6, fn="vt_gemm_f32"):
    import torch
    from vit_amd._lib import check, lib
    rng = np.random.default_rng(seed)
    ar, ac = (K, M) if ta else (M, K)
    br, bc = (N, K) if tb else (K, N)
    lda, ldb, ldc = ac + pad, bc + pad, N + pad
    A = rng.standard_normal((batch, ar, lda)).astype(np.float32)
    B = rng.standard_normal((batch, br, ldb)).astype(np.float32)
    C0 = rng.standard_normal((batch, M, ldc)).astype(np.float32)
    bv = rng.standard_normal(N).astype(np.float32) if bias else None
    dA, dB, dC = (torch.from_numpy(t).cuda() for t in (A, B, C0))
    dbias = torch.from_numpy(bv).cuda() if bias else None
    ws = torch.zeros(max(ws_floats, 1), dtype=torch.float32, device="cuda")
    s = torch.cuda.current_stream().cuda_stream
    check(getattr(lib, fn)(int(ta), int(tb), M, N, K, alpha, C.c_void_p(dA.data_ptr()), lda, ar * lda,
                           C.c_void_p(dB.data_ptr()), ldb, br * ldb, int(acc), C.c_void_p(dC.data_ptr()), ldc, M * ldc,
                           C.c_void_p(dbias.data_ptr()) if bias else None, batch,
                           C.c_void_p(ws.data_ptr()) if ws_floats else None, ws_floats, C.c_void_p(s)), fn)
    got = dC.cpu().numpy()
    a = A[:, :, :ac].astype(np.float64)
    b = B[:, :, :bc].astype(np.float64)
    if ta:
        a = a.transpose(0, 2, 1)
    if tb:
        b = b.transpose(0, 2, 1)
    want = alpha * np.matmul(a, b)
    if bias:
        want = want + bv.astype(np.float64)
    if acc:
        want = want + C0[:, :, :N]
    assert rel_err(got[:, :, :N], want) < tol, (ta, tb, M, N, K)
    if pad:
        assert np.array_equal(got[:, :, N:], C0[:, :, N:])  # nothing written outside the matrix


@pytest.mark.parametrize("ta", [0, 1])
@pytest.mark.parametrize("tb", [0, 1])
def test_gemm_transposes_and_edges(ta, tb):
    _run(ta, tb, 64, 64, 16)
    _run(ta, tb, 1, 1, 1)
    _run(ta, tb, 130, 75, 37, pad=3, seed=1)
    _run(ta, tb, 100, 16, 200, batch=5, alpha=0.25, bias=True, seed=2)
    _run(ta, tb, 33, 129, 64, batch=3, acc=True, pad=1, seed=3)
    _run(ta, tb, 6000, 64, 64, seed=4)  # tall and narrow (row-panel kernel when A is not transposed)
    _run(ta, tb, 2500, 16, 128, batch=3, alpha=0.5, bias=True, acc=True, seed=5)
    _run(ta, tb, 2100, 128, 20, pad=4, seed=6)


def test_gemm_sliced_contraction():
    # weight-gradient shape: X^T . dY over 100 352 rows, result 64 x 128
    _run(1, 0, 64, 128, 100352, ws_floats=1 << 22, seed=4)
    _run(1, 0, 64, 16, 50000, batch=2, bias=True, acc=True, ws_floats=1 << 20, seed=5)
    _run(1, 0, 27, 64, 9001, ws_floats=1 << 20, seed=6)
    # without a workspace the same product runs unsliced: 50 000 sequential fp32 additions per element
    _run(1, 0, 64, 16, 50000, seed=7, tol=5e-5)


@pytest.mark.parametrize("ta", [0, 1])
@pytest.mark.parametrize("tb", [0, 1])
def test_gemm_split_products_keep_fp32_accuracy(ta, tb):
    """vt_gemm_f32x3 (operands as bf16 hi + lo, three partial products, fp32 accumulation): every case of the exact
    kernel's test, against numpy in double precision at 2e-5 of the largest element (2^-17 per product, random in
    sign, over sums of Gaussian terms), including a 512-deep product of the wide configuration's shape."""
    kw = dict(fn="vt_gemm_f32x3", tol=2e-5)
    _run(ta, tb, 64, 64, 16, **kw)
    _run(ta, tb, 1, 1, 1, **kw)
    _run(ta, tb, 130, 75, 37, pad=3, seed=1, **kw)
    _run(ta, tb, 100, 16, 200, batch=5, alpha=0.25, bias=True, seed=2, **kw)
    _run(ta, tb, 33, 129, 64, batch=3, acc=True, pad=1, seed=3, **kw)
    _run(ta, tb, 3136, 512, 512, seed=8, **kw)
    _run(ta, tb, 256, 196, 512, batch=4, seed=9, **kw)
    _run(ta, tb, 6000, 64, 64, seed=10, **kw)  # tall and narrow (row-panel kernel when A is not transposed)
    _run(ta, tb, 2500, 16, 128, batch=3, alpha=0.5, bias=True, acc=True, seed=11, **kw)


def _sum_cases(form, tol, cases):
    import torch
    from vit_amd._lib import VtGemmTerm, check, lib
    rng = np.random.default_rng(11)
    for (M, N, batch, ks, tbs, shared, bias, acc) in cases:
        terms = (VtGemmTerm * len(ks))()
        keep = []
        want = np.zeros((batch, M, N))
        for t, (K, tb, sh) in enumerate(zip(ks, tbs, shared)):
            A = rng.standard_normal((batch, M, K)).astype(np.float32)
            Bshape = (1 if sh else batch,) + ((N, K) if tb else (K, N))
            B = rng.standard_normal(Bshape).astype(np.float32)
            dA, dB = torch.from_numpy(A).cuda(), torch.from_numpy(B).cuda()
            keep += [dA, dB]
            terms[t] = VtGemmTerm(dA.data_ptr(), dB.data_ptr(), M * K, 0 if sh else N * K, K, K, K if tb else N, tb)
            B64 = B.astype(np.float64)
            want += A.astype(np.float64) @ (np.swapaxes(B64, 1, 2) if tb else B64)
        want *= 0.5
        c0 = rng.standard_normal((batch, M, N)).astype(np.float32)
        bv = rng.standard_normal(N).astype(np.float32)
        if bias:
            want += bv
        if acc:
            want += c0
        dC, dbias = torch.from_numpy(c0.copy()).cuda(), torch.from_numpy(bv).cuda()
        check(lib.vt_gemm_sum(form, len(ks), terms, M, N, 0.5, int(acc), C.c_void_p(dC.data_ptr()), N, M * N,
                              C.c_void_p(dbias.data_ptr()) if bias else None, batch,
                              C.c_void_p(torch.cuda.current_stream().cuda_stream)), "vt_gemm_sum")
        torch.cuda.synchronize()
        assert rel_err(dC.cpu().numpy(), want) < tol, (M, N, ks)


@pytest.mark.parametrize("form,tol", [(0, 2e-6), (2, 2e-5)])
def test_gemm_sum_of_products(form, tol):
    """vt_gemm_sum: up to four A_t . op(B_t) terms accumulated in registers into one C (bias, accumulate, per-term
    transpose of B, operands shared by the batch through stride 0, ragged sizes) against numpy in double precision."""
    from vit_amd._lib import VtGemmTerm, lib
    _sum_cases(form, tol, [
        (200, 64, 3, (128, 64, 16, 16), (1, 1, 0, 1), (1, 1, 0, 1), False, False),  # the input-delta sum's shape
        (300, 128, 2, (64, 16), (0, 0), (1, 0), True, False),                       # the projector output's
        (77, 45, 2, (5, 33, 18), (0, 1, 0), (0, 0, 1), True, True),
        (130, 200, 1, (300,), (1,), (0,), False, True),
        (256, 256, 4, (256, 196), (0, 0), (1, 0), True, False),
    ])
    _sum_cases(form, tol, [
        # tall and narrow: the row-panel kernel (B resident in LDS, A straight into operand layout)
        (5000, 64, 2, (128, 64, 16, 16), (1, 1, 0, 1), (1, 1, 0, 1), False, False),
        (4100, 128, 3, (64, 16), (0, 0), (1, 0), True, False),
        (2049, 45, 2, (36, 20), (0, 1), (0, 1), True, True),
        (70000, 16, 1, (64,), (0,), (1,), True, False),
        (3000, 30, 2, (8,), (1,), (0,), False, True),
        (4100, 196, 2, (64,), (0,), (0,), True, False),      # thirteen result tiles
        (2300, 128, 2, (64, 196), (0, 0), (1, 0), True, False),  # the wide projector output: more than 64 KB of LDS
        (2300, 200, 1, (128, 36), (1, 0), (0, 1), False, True),
    ])
    _sum_cases(form, tol, [
        # deep, narrow, many rows, op(B) beyond a compute unit's LDS: the tiled kernel with 128-row tiles
        (2100, 64, 128, (208, 208, 208), (1, 0, 1), (1, 1, 0), True, False),
    ])
    # argument checks
    bad = (VtGemmTerm * 1)()
    import torch
    dC = torch.zeros(16, device="cuda")
    assert lib.vt_gemm_sum(0, 5, bad, 4, 4, 1.0, 0, C.c_void_p(dC.data_ptr()), 4, 16, None, 1, None) != 0
    assert lib.vt_gemm_sum(0, 1, bad, 4, 4, 1.0, 0, C.c_void_p(dC.data_ptr()), 4, 16, None, 1, None) != 0  # NULL operands


@pytest.mark.parametrize("form,tol", [(0, 3e-6), (2, 3e-5)])
def test_gemm_with_softmax_epilogues(form, tol):
    """vt_gemm_softmax_rows: product + softmax over each row, and product + softmax backward against a, where the
    row-panel kernel applies the epilogue to its accumulators (tall M; N <= 16 or 129 ... 208) and where it falls back
    to two launches (short M, other N), against numpy in double precision."""
    import torch
    from vit_amd._lib import check, lib
    rng = np.random.default_rng(3)
    for (M, N, K, batch, tb) in [(5000, 16, 64, 2, 1), (3000, 10, 20, 3, 0), (2048, 16, 128, 1, 1), (70000, 7, 36, 1, 0),
                                 (300, 16, 64, 2, 1), (2500, 33, 64, 2, 0),
                                 (4100, 196, 64, 2, 0), (2100, 196, 128, 2, 1), (2600, 150, 40, 1, 0)]:  # thirteen-tile rows
        A = rng.standard_normal((batch, M, K)).astype(np.float32)
        B = (rng.standard_normal((batch, N, K) if tb else (batch, K, N)) * 0.3).astype(np.float32)
        prod = 0.5 * (A.astype(np.float64) @ (np.swapaxes(B, 1, 2) if tb else B).astype(np.float64))
        e = np.exp(prod - prod.max(axis=2, keepdims=True))
        soft = e / e.sum(axis=2, keepdims=True)
        a = rng.random((batch, M, N)).astype(np.float32)
        bwd = a.astype(np.float64) * (prod - (a.astype(np.float64) * prod).sum(axis=2, keepdims=True))
        dA, dB, da = torch.from_numpy(A).cuda(), torch.from_numpy(B).cuda(), torch.from_numpy(a).cuda()
        for epi, want in ((1, soft), (2, bwd)):
            dC = torch.full((batch, M, N), 7.0, dtype=torch.float32, device="cuda")
            check(lib.vt_gemm_softmax_rows(form, epi, C.c_void_p(da.data_ptr()) if epi == 2 else None, tb, M, N, K, 0.5,
                                           C.c_void_p(dA.data_ptr()), K, M * K, C.c_void_p(dB.data_ptr()), K if tb else N,
                                           N * K, C.c_void_p(dC.data_ptr()), M * N, batch,
                                           C.c_void_p(torch.cuda.current_stream().cuda_stream)), "vt_gemm_softmax_rows")
            torch.cuda.synchronize()
            assert rel_err(dC.cpu().numpy(), want) < tol, (M, N, K, epi)
    assert lib.vt_gemm_softmax_rows(1, 1, None, 0, 4, 4, 4, 1.0, C.c_void_p(dA.data_ptr()), 4, 0, C.c_void_p(dB.data_ptr()), 4, 0,
                                    C.c_void_p(dC.data_ptr()), 16, 1, None) != 0  # no bf16 form
    assert lib.vt_gemm_softmax_rows(0, 2, None, 0, 4, 4, 4, 1.0, C.c_void_p(dA.data_ptr()), 4, 0, C.c_void_p(dB.data_ptr()), 4, 0,
                                    C.c_void_p(dC.data_ptr()), 16, 1, None) != 0  # backward needs a


def test_gemm_sum_with_batchnorm_partial_records():
    """vt_gemm_sum_colstats: the sum of products plus, per block of rows, (sum, squared distances from the block's mean)
    of every result column; vt_bn_stats_finalize on those records gives the statistics of the result (numpy, float64)."""
    import torch
    from vit_amd._lib import VtGemmTerm, check, lib
    rng = np.random.default_rng(11)
    for (form, M, N, batch, ks) in [(2, 4096, 128, 3, (64, 16)), (2, 5000, 100, 1, (64, 196)), (2, 2048, 128, 16, (32,)),
                                    (0, 70000, 64, 1, (28,)), (0, 4096, 40, 2, (12, 8))]:
        terms = (VtGemmTerm * len(ks))()
        keep, want = [], np.zeros((batch, M, N))
        for t, K in enumerate(ks):
            A = (rng.standard_normal((batch, M, K)) + 0.3).astype(np.float32)
            B = rng.standard_normal((batch, K, N)).astype(np.float32)
            dA, dB = torch.from_numpy(A).cuda(), torch.from_numpy(B).cuda()
            keep += [dA, dB]
            terms[t] = VtGemmTerm(dA.data_ptr(), dB.data_ptr(), M * K, K * N, K, K, N, 0)
            want += A.astype(np.float64) @ B.astype(np.float64)
        bias = (rng.standard_normal(N) * 3).astype(np.float32)
        want += bias
        pivot = (want.reshape(-1, N).mean(0) + rng.standard_normal(N) * 0.5).astype(np.float32)
        dbias, dpiv = torch.from_numpy(bias).cuda(), torch.from_numpy(pivot).cuda()
        dC = torch.zeros((batch, M, N), dtype=torch.float32, device="cuda")
        dC2 = torch.zeros_like(dC)
        part = torch.full((batch * ((M + 127) // 128) * 2 * N,), -3.0, dtype=torch.float32, device="cuda")
        rpb = C.c_int(0)
        s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        check(lib.vt_gemm_sum_colstats(form, len(ks), terms, M, N, 1.0, C.c_void_p(dC.data_ptr()), N, M * N,
                                       C.c_void_p(dbias.data_ptr()), batch, C.c_void_p(dpiv.data_ptr()),
                                       C.c_void_p(part.data_ptr()), part.numel(), C.byref(rpb), s), "vt_gemm_sum_colstats")
        check(lib.vt_gemm_sum(form, len(ks), terms, M, N, 1.0, 0, C.c_void_p(dC2.data_ptr()), N, M * N,
                              C.c_void_p(dbias.data_ptr()), batch, s), "vt_gemm_sum")
        torch.cuda.synchronize()
        assert torch.equal(dC, dC2)
        rows = rpb.value
        assert rows >= 128 and (batch == 1 or M % rows == 0)
        nb = batch * ((M + rows - 1) // rows)
        got = dC.cpu().numpy().astype(np.float64).reshape(batch * M, N)
        rec = part.cpu().numpy()[:nb * 2 * N].reshape(nb, 2, N)
        for b in (0, nb // 2, nb - 1):
            blk = got[b * rows:(b + 1) * rows] if batch == 1 else got.reshape(batch, M, N)[b // (M // rows)][(b % (M // rows)) * rows:][:rows]
            assert rel_err(rec[b, 0], blk.sum(0)) < 2e-6, (M, N, b)
            assert rel_err(rec[b, 1], ((blk - blk.mean(0)) ** 2).sum(0)) < 2e-5, (M, N, b)
        stat = torch.zeros(2 * N, dtype=torch.float32, device="cuda")
        rm, rv = torch.zeros(N, device="cuda"), torch.ones(N, device="cuda")
        check(lib.vt_bn_stats_finalize(C.c_void_p(part.data_ptr()), nb, N, rows, batch * M, C.c_void_p(rm.data_ptr()),
                                       C.c_void_p(rv.data_ptr()), C.c_void_p(stat.data_ptr()), s), "vt_bn_stats_finalize")
        torch.cuda.synchronize()
        st = stat.cpu().numpy()
        assert rel_err(st[:N], got.mean(0)) < 2e-6
        assert rel_err(st[N:], 1.0 / np.sqrt(got.var(0) + 1e-5)) < 2e-5
    # shapes the row-panel kernel does not take with records: refused, nothing launched
    assert lib.vt_gemm_sum_colstats(form, len(ks), terms, 300, N, 1.0, C.c_void_p(dC.data_ptr()), N, 300 * N, None, 1, None,
                                    C.c_void_p(part.data_ptr()), part.numel(), C.byref(rpb), s) == -2  # too few rows


def _bf16_round(a):
    u = a.astype(np.float32).view(np.uint32).astype(np.uint64)
    u = (u + 0x7FFF + ((u >> 16) & 1)) >> 16 << 16
    return u.astype(np.uint32).view(np.float32)


@pytest.mark.parametrize("ta,tb", [(0, 0), (0, 1), (1, 0), (1, 1)])
def test_gemm_bf16_operands(ta, tb):
    """vt_gemm_bf16_mfma = the fp32 product of the bf16-rounded operands (fp32 accumulation): exact
    to accumulation order against numpy on rounded inputs, within 1e-2 of the unrounded product."""
    import torch
    from vit_amd._lib import check, lib
    rng = np.random.default_rng(11 + 2 * ta + tb)
    for (M, N, K, batch, pad) in ((64, 64, 64, 1, 0), (130, 75, 37, 2, 3), (100, 196, 512, 3, 0)):
        ar, ac = (K, M) if ta else (M, K)
        br, bc = (N, K) if tb else (K, N)
        lda, ldb, ldc = ac + pad, bc + pad, N + pad
        A = rng.standard_normal((batch, ar, lda)).astype(np.float32)
        B = rng.standard_normal((batch, br, ldb)).astype(np.float32)
        bias = rng.standard_normal(N).astype(np.float32)
        dA, dB = torch.from_numpy(A).cuda(), torch.from_numpy(B).cuda()
        dC = torch.zeros(batch, M, ldc, device="cuda")
        db = torch.from_numpy(bias).cuda()
        check(lib.vt_gemm_bf16_mfma(ta, tb, M, N, K, 0.5, C.c_void_p(dA.data_ptr()), lda, ar * lda, C.c_void_p(dB.data_ptr()),
                                    ldb, br * ldb, 0, C.c_void_p(dC.data_ptr()), ldc, M * ldc, C.c_void_p(db.data_ptr()), batch,
                                    None, 0, C.c_void_p(torch.cuda.current_stream().cuda_stream)), "vt_gemm_bf16_mfma")
        got = dC.cpu().numpy()[:, :, :N]

        def prod(a, b):
            a, b = a[:, :, :ac].astype(np.float64), b[:, :, :bc].astype(np.float64)
            a = a.transpose(0, 2, 1) if ta else a
            b = b.transpose(0, 2, 1) if tb else b
            return 0.5 * np.matmul(a, b) + bias
        assert rel_err(got, prod(_bf16_round(A), _bf16_round(B))) < 2e-6, (M, N, K)
        assert rel_err(got, prod(A, B)) < 1e-2, (M, N, K)


def test_conv3x3_rows_gathered_from_the_images():
    """vt_conv3x3_rows_colstats: the im2col matrix it writes, the product and the partial records against numpy, and
    the product bit for bit against vt_gemm_sum_colstats on that matrix (same operands, same instruction order)."""
    import torch
    from vit_amd._lib import VtGemmTerm, check, lib
    rng = np.random.default_rng(23)
    for (n, Ci, H, Co, ld) in [(3, 3, 48, 64, 28), (2, 1, 32, 40, 12), (9, 2, 16, 64, 20)]:
        X = rng.standard_normal((n, Ci, H, H)).astype(np.float32)
        Wc = rng.standard_normal((Co, Ci * 9)).astype(np.float32)
        Wp = np.zeros((Co, ld), np.float32)
        Wp[:, :Ci * 9] = Wc
        bias = rng.standard_normal(Co).astype(np.float32)
        M = n * H * H
        Xp = np.pad(X, ((0, 0), (0, 0), (1, 1), (1, 1)))
        col = np.zeros((n, H, H, ld), np.float32)
        for ci in range(Ci):
            for kh in range(3):
                for kw in range(3):
                    col[..., (ci * 3 + kh) * 3 + kw] = Xp[:, ci, kh:kh + H, kw:kw + H]
        col = col.reshape(M, ld)
        want = col.astype(np.float64) @ Wp.T.astype(np.float64) + bias
        pivot = (want.mean(0) + 0.1).astype(np.float32)
        dX, dW, db, dp = (torch.from_numpy(a).cuda() for a in (X, Wp, bias, pivot))
        dcol = torch.full((M, ld), 7.0, dtype=torch.float32, device="cuda")
        dC = torch.zeros((M, Co), dtype=torch.float32, device="cuda")
        dC2 = torch.zeros_like(dC)
        part = torch.zeros(((M + 127) // 128) * 2 * Co, dtype=torch.float32, device="cuda")
        part2 = torch.zeros_like(part)
        rpb, rpb2 = C.c_int(0), C.c_int(0)
        s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        check(lib.vt_conv3x3_rows_colstats(C.c_void_p(dX.data_ptr()), n, Ci, H, C.c_void_p(dW.data_ptr()), ld, Co,
                                           C.c_void_p(db.data_ptr()), C.c_void_p(dcol.data_ptr()), C.c_void_p(dC.data_ptr()),
                                           C.c_void_p(dp.data_ptr()), C.c_void_p(part.data_ptr()), part.numel(), C.byref(rpb), s),
              "vt_conv3x3_rows_colstats")
        torch.cuda.synchronize()
        assert np.array_equal(dcol.cpu().numpy(), col), (n, Ci, H)
        got = dC.cpu().numpy()
        assert rel_err(got, want) < 1e-5
        term = (VtGemmTerm * 1)()
        term[0] = VtGemmTerm(dcol.data_ptr(), dW.data_ptr(), 0, 0, ld, ld, ld, 1)
        check(lib.vt_gemm_sum_colstats(0, 1, term, M, Co, 1.0, C.c_void_p(dC2.data_ptr()), Co, 0, C.c_void_p(db.data_ptr()), 1,
                                       C.c_void_p(dp.data_ptr()), C.c_void_p(part2.data_ptr()), part2.numel(), C.byref(rpb2), s),
              "vt_gemm_sum_colstats")
        torch.cuda.synchronize()
        assert rpb.value == rpb2.value and torch.equal(dC, dC2) and torch.equal(part, part2)
        rows = rpb.value
        nb = (M + rows - 1) // rows
        rec = part.cpu().numpy()[:nb * 2 * Co].reshape(nb, 2, Co)
        blk = got[(nb - 1) * rows:].astype(np.float64)
        assert rel_err(rec[nb - 1, 0], blk.sum(0)) < 2e-6
        assert rel_err(rec[nb - 1, 1], ((blk - blk.mean(0)) ** 2).sum(0)) < 2e-5
    # image sides that are not a multiple of 16: refused, nothing launched
    assert lib.vt_conv3x3_rows_colstats(C.c_void_p(dX.data_ptr()), 1, 1, 28, C.c_void_p(dW.data_ptr()), 12, Co,
                                        None, C.c_void_p(dcol.data_ptr()), C.c_void_p(dC.data_ptr()), None,
                                        C.c_void_p(part.data_ptr()), part.numel(), C.byref(rpb), s) == -2
