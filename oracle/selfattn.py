"""TEST INFRASTRUCTURE ONLY -- never imported by the product path.

CPU restatement (numpy, fp32 storage, fp32 accumulation order of a plain row-by-column product) of the
reference's SelfAttention layer, layers/self_attention.hpp.  Pinned against fixtures made by the real
layer (tests/golden/selfattn_*.npz, tools/make_golden_selfattn.py) in tests/test_oracle_selfattn.py."""
import numpy as np


def _softmax_last(s):
    # Tensor::softmax(dim): subtract the maximum, exp, sum, divide (tensor_cpu.hpp:894-944)
    e = np.exp(s - s.max(axis=-1, keepdims=True), dtype=np.float32)
    return (e / e.sum(axis=-1, keepdims=True, dtype=np.float32)).astype(np.float32)


def forward(x, wk, wq):
    """self_attention.hpp:31-55 -> (out, key, query, attn)."""
    B, T, D = x.shape
    xf = x.reshape(B * T, D)
    key = (xf @ wk).reshape(B, T, -1)                      # key_linear->forward, no bias (:42)
    query = (xf @ wq).reshape(B, T, -1)                    # query_linear->forward (:43)
    attn = _softmax_last(key @ query.transpose(0, 2, 1))   # key.matmul(query_T).softmax(2) (:51)
    return (attn @ x).astype(np.float32), key, query, attn  # attn.matmul(x) (:52)


def backward(x, wk, wq, delta):
    """self_attention.hpp:57-136 -> (g_wk, g_wq, dx)."""
    B, T, D = x.shape
    _, key, query, attn = forward(x, wk, wq)                # recomputed as the reference does (:81-89)
    d_attn = delta @ x.transpose(0, 2, 1)                   # :99-100
    dx = attn.transpose(0, 2, 1) @ delta                    # :103-104
    ds = attn * (d_attn - (d_attn * attn).sum(-1, keepdims=True))  # Tensor::softmax_backward (tensor_cpu.hpp:946-950)
    d_key = ds @ query                                      # :113
    d_query = (key.transpose(0, 2, 1) @ ds).transpose(0, 2, 1)  # :117-119
    xf = x.reshape(B * T, D)
    dkf, dqf = d_key.reshape(B * T, -1), d_query.reshape(B * T, -1)
    g_wk, g_wq = xf.T @ dkf, xf.T @ dqf                     # DenseLayer::backward (dense_layer.hpp:92-93)
    dx = dx + (dkf @ wk.T).reshape(B, T, D) + (dqf @ wq.T).reshape(B, T, D)  # :129-135
    return g_wk.astype(np.float32), g_wq.astype(np.float32), dx.astype(np.float32)
