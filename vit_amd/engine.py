"""Host-side mirror of the reference's Model / Trainer / DataLoader interface on top of the C ABI.

Names follow the reference: `VTCNN` (model/vit_cnn.hpp), `Trainer` (utils/trainer.hpp),
`DataLoader` / `load_dataset` (utils/load_dataset.hpp).  PyTorch is used only for device memory,
streams and (optionally) torch.distributed; every computation is a HIP kernel in libvtcnn_hip.so.
"""
import ctypes as C
import math
import os
import time

import numpy as np
import torch

from ._lib import VtDims, VtLayout, check, lib


def Dims(num_classes=10, image_channels=1, input_dim=28, tokens=16, token_channels=32, vt_channels=16, heads=2):
    """Hyper-parameters with the reference's defaults (main.cpp:66-78)."""
    return VtDims(image_channels, input_dim, tokens, token_channels, vt_channels, heads, num_classes)


def _stream_handle(device=None):
    """The caller's current stream ON THE GIVEN DEVICE (a model's own), not on whichever device is current."""
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


class Layout:
    def __init__(self, lo):
        self.n_trainable = lo.n_trainable
        self.n_total = lo.n_total
        self.n_trainable_exact = lo.n_trainable_exact
        self.tensors = {}
        self.order = []
        for k in range(lo.n_tensors):
            t = lo.t[k]
            name = t.name.decode()
            self.tensors[name] = (t.offset, t.size, tuple(t.shape[:t.ndim]), t.kind)
            self.order.append(name)

    def view(self, arena, name):
        off, size, shape, _ = self.tensors[name]
        return arena[off:off + size].reshape(shape)


def layout_sizes(d):
    """(n_total, n_trainable) of the parameter arena for hyper-parameters d; mirrors include/vtcnn_layout.h
    (every tensor padded to a multiple of 4 floats)."""
    c2, c1, L, D, h, K, ci = d.vt_channels, d.vt_channels // 2, d.tokens, d.token_channels, d.heads, d.num_classes, d.image_channels
    pad = lambda n: (n + 3) & ~3  # noqa: E731
    hd = D // h
    train = [c1 * ci * 9, c1, c1, c1] + [D * hd] * (3 * h) + [D * D, D * D, D, D * D, D, D, D, D, D,
                                                               c1 * D, D * D, D * c2, c2, c2, c2, c1 * c2, c2,
                                                               c2 * 196 * K, K]
    rest = [c1 * L, L, c1 * D, D, c1, c1, D, D, D, D, c2, c2]
    n_train = sum(pad(x) for x in train)
    return n_train + sum(pad(x) for x in rest), n_train


class VTCNN:
    """VTCNN model on one MI355X (reference model/vit_cnn.hpp:14-557).

    forward / backward / update_weights / zero_grad / set_training / save / load keep the
    reference's meaning.  Inputs are torch CUDA tensors ([N,Ci,H,W] float32 images, int32 labels);
    the parameter and gradient arenas are torch tensors so that torch.distributed (RCCL) can
    all-reduce gradients in place.
    """

    def __init__(self, dims=None, max_batch=32, device=None, seed=42, lr=0.001):
        if not torch.cuda.is_available():
            raise RuntimeError("vit_amd.VTCNN needs a HIP device; there is no CPU path")
        self.dims = dims or Dims()
        self.device = torch.device("cuda", torch.cuda.current_device() if device is None else device)
        self.max_batch = int(max_batch)
        self.lr = float(lr)
        lo = VtLayout()
        self._h = C.c_void_p()
        with torch.cuda.device(self.device):
            # arenas are torch tensors so that collectives can address them in place
            n_total, n_train = self._layout_sizes()
            self.arena = torch.zeros(n_total, dtype=torch.float32, device=self.device)
            self.grads = torch.zeros(n_train, dtype=torch.float32, device=self.device)
            check(lib.vt_model_create(C.byref(self.dims), self.max_batch, self.device.index,
                                      C.c_void_p(self.arena.data_ptr()), C.c_void_p(self.grads.data_ptr()),
                                      C.byref(self._h)), "vt_model_create")
        check(lib.vt_model_layout(self._h, C.byref(lo)), "vt_model_layout")
        self.layout = Layout(lo)
        self.fused = lib.vt_model_is_fused(self._h) == 1  # headline shape: fused kernels; otherwise the general step
        self.training = True
        self._probs = torch.zeros(self.max_batch, self.dims.num_classes, dtype=torch.float32, device=self.device)
        self._n = 0
        if seed is not None:
            check(lib.vt_model_init_seed(self._h, int(seed)), "vt_model_init_seed")

    def _layout_sizes(self):
        return layout_sizes(self.dims)

    def __del__(self):
        h = getattr(self, "_h", None)
        if h:
            lib.vt_model_destroy(h)
            self._h = None

    # ---- reference Model API -------------------------------------------------------------
    def set_training(self, is_training):
        self.training = bool(is_training)
        check(lib.vt_model_set_training(self._h, int(self.training)))

    def set_concurrency(self, on):
        """Issue the two independent halves of the backward on two streams (default) or on one
        (vt_model_set_concurrency); same bytes either way."""
        check(lib.vt_model_set_concurrency(self._h, int(bool(on))), "vt_model_set_concurrency")

    def set_fused_reductions(self, on):
        """BatchNorm reductions closed inside their producer kernels (opt-in; measured no faster) or by separate
        finalize launches, the default (vt_model_set_fused_reductions)."""
        check(lib.vt_model_set_fused_reductions(self._h, 1 if on else 0), "vt_model_set_fused_reductions")

    def set_matmul_precision(self, bf16):
        """General-shape models only: route the forward-type products through bf16 operands
        (vt_model_set_matmul_precision); parity drops from 1e-4 to 1e-2."""
        check(lib.vt_model_set_matmul_precision(self._h, int(bool(bf16))), "vt_model_set_matmul_precision")

    def keep_internals(self, on=True):
        """Test aid: also store the attention maps "fwd.tok_A" / "fwd.proj_a" for read()."""
        check(lib.vt_model_keep_internals(self._h, int(bool(on))))

    def forward(self, x):
        """VTCNN::forward: x [N,Ci,H,W] float32 CUDA tensor -> softmax probabilities [N,K]."""
        assert x.is_cuda and x.dtype == torch.float32 and x.is_contiguous()
        n = x.shape[0]
        d = self.dims
        if tuple(x.shape[1:]) != (d.image_channels, d.input_dim, d.input_dim):
            raise ValueError("input shape %s does not match the model" % (tuple(x.shape),))
        if n < 1 or n > self.max_batch:
            raise ValueError("batch %d outside [1, max_batch=%d]" % (n, self.max_batch))
        self._n = n
        self._x = x  # keep alive: backward re-reads the images
        check(lib.vt_model_forward(self._h, C.c_void_p(x.data_ptr()), n, C.c_void_p(self._probs.data_ptr()),
                                   _stream_handle(self.device)), "vt_model_forward")
        return self._probs[:n]

    def train_step(self, x, labels, batch_size=32.0):
        """forward + backward + update_weights in one call (vt_model_train_step): the same bytes as the three
        calls, one launch fewer on the headline shape.  Returns the softmax probabilities [N,K]."""
        assert x.is_cuda and x.dtype == torch.float32 and x.is_contiguous()
        assert labels.is_cuda and labels.dtype == torch.int32 and labels.is_contiguous()
        n = x.shape[0]
        d = self.dims
        if tuple(x.shape[1:]) != (d.image_channels, d.input_dim, d.input_dim) or labels.numel() != n:
            raise ValueError("input shape %s / labels %s do not match the model" % (tuple(x.shape), tuple(labels.shape)))
        if n < 1 or n > self.max_batch:
            raise ValueError("batch %d outside [1, max_batch=%d]" % (n, self.max_batch))
        self._n = n
        self._x = x
        check(lib.vt_model_train_step(self._h, C.c_void_p(x.data_ptr()), n, C.c_void_p(labels.data_ptr()), self.lr,
                                      float(batch_size), C.c_void_p(self._probs.data_ptr()), _stream_handle(self.device)),
              "vt_model_train_step")
        return self._probs[:n]

    def train_step_grouped(self, x, labels, group, batch_size=32.0):
        """n / group independent reference mini-batches of `group` images trained side by side in one launch set
        (vt_model_train_step_grouped): per-mini-batch statistics and attention groups, gradients summed, ONE update with
        divisor (n / group) * batch_size.  Returns the softmax probabilities [N,K]."""
        assert x.is_cuda and x.dtype == torch.float32 and x.is_contiguous()
        assert labels.is_cuda and labels.dtype == torch.int32 and labels.is_contiguous()
        n = x.shape[0]
        d = self.dims
        if tuple(x.shape[1:]) != (d.image_channels, d.input_dim, d.input_dim) or labels.numel() != n:
            raise ValueError("input shape %s / labels %s do not match the model" % (tuple(x.shape), tuple(labels.shape)))
        if n < 1 or n > self.max_batch or n % int(group) != 0:
            raise ValueError("batch %d is not a multiple of group %d inside [1, max_batch=%d]" % (n, group, self.max_batch))
        self._n = n
        self._x = x
        divisor = float(batch_size) * (n // int(group))
        check(lib.vt_model_train_step_grouped(self._h, C.c_void_p(x.data_ptr()), n, int(group), C.c_void_p(labels.data_ptr()),
                                              self.lr, divisor, C.c_void_p(self._probs.data_ptr()), _stream_handle(self.device)),
              "vt_model_train_step_grouped")
        return self._probs[:n]

    def forward_grouped(self, x, group):
        """Forward of consecutive mini-batches of `group` images in one launch set, each with its own batch statistics
        (vt_model_forward_grouped): an evaluation sweep in eval mode; in training mode the forward half of a grouped
        training step (backward() afterwards leaves the sum of the mini-batches' gradients)."""
        assert x.is_cuda and x.dtype == torch.float32 and x.is_contiguous()
        n = x.shape[0]
        if n < 1 or n > self.max_batch:
            raise ValueError("batch %d outside [1, max_batch=%d]" % (n, self.max_batch))
        self._n = n
        self._x = x
        check(lib.vt_model_forward_grouped(self._h, C.c_void_p(x.data_ptr()), n, int(group),
                                           C.c_void_p(self._probs.data_ptr()), _stream_handle(self.device)), "vt_model_forward_grouped")
        return self._probs[:n]

    def backward(self, labels):
        """VTCNN::backward for the last forward; labels int32 CUDA tensor [N] (class indices; the
        reference passes the one-hot matrix, utils/trainer.hpp:47-52)."""
        assert labels.is_cuda and labels.dtype == torch.int32 and labels.numel() == self._n
        check(lib.vt_model_backward(self._h, C.c_void_p(labels.data_ptr()), _stream_handle(self.device)), "vt_model_backward")

    def backward_main(self, labels):
        """Backward of everything but the attention block: afterwards every gradient outside
        attention_range() is final (vt_model_backward_main)."""
        assert labels.is_cuda and labels.dtype == torch.int32 and labels.numel() == self._n
        check(lib.vt_model_backward_main(self._h, C.c_void_p(labels.data_ptr()), _stream_handle(self.device)), "vt_model_backward_main")

    def backward_attention(self, out=None):
        """The attention block's backward: writes the gradients inside attention_range(), or into
        `out` (float32 CUDA tensor of that many elements) instead of the gradient arena."""
        if out is not None:
            assert out.is_cuda and out.dtype == torch.float32 and out.is_contiguous() and out.numel() == self.attention_range()[1]
        check(lib.vt_model_backward_attention(self._h, C.c_void_p(out.data_ptr() if out is not None else None), _stream_handle(self.device)),
              "vt_model_backward_attention")

    def update_block(self, batch_size, first, grads):
        """SGD on arena[first : first + len(grads)] with the gradients in `grads` (cleared afterwards)."""
        assert grads.is_cuda and grads.dtype == torch.float32 and grads.is_contiguous()
        check(lib.vt_sgd_update(C.c_void_p(self.arena.data_ptr() + 4 * int(first)), C.c_void_p(grads.data_ptr()), grads.numel(),
                                self.lr, float(batch_size), 1, _stream_handle(self.device)), "vt_sgd_update")

    def attention_range(self):
        """(first, count): the floats of the arenas that hold mha.q*/k*/v*."""
        first, count = C.c_int(0), C.c_int(0)
        check(lib.vt_model_attention_range(self._h, C.byref(first), C.byref(count)), "vt_model_attention_range")
        return first.value, count.value

    def update_weights(self, batch_size=32.0):
        """VTCNN::update_weights + zero_grad with the reference's constant divisor (trainer.hpp:53)."""
        check(lib.vt_model_sgd_step(self._h, self.lr, float(batch_size), _stream_handle(self.device)), "vt_model_sgd_step")

    def update_weights_adam(self, batch_size=32.0, lr=None, beta1=0.9, beta2=0.999, eps=1e-8):
        """One fused Adam pass over the trainable region (extension: the reference has SGD only)."""
        check(lib.vt_model_adam_step(self._h, self.lr if lr is None else float(lr), beta1, beta2, eps, float(batch_size),
                                     _stream_handle(self.device)), "vt_model_adam_step")

    def update_weights_range(self, batch_size, first, count):
        """The same update over floats [first, first + count) of the trainable region."""
        check(lib.vt_model_sgd_step_range(self._h, self.lr, float(batch_size), int(first), int(count), _stream_handle(self.device)),
              "vt_model_sgd_step_range")

    def zero_grad(self):
        with self._settled():
            self.grads.zero_()

    def capture_step(self, x, labels, batch_size=32.0):
        """Captures forward + backward + SGD on the fixed device buffers x / labels as one HIP
        graph; replay with step_graph() after refilling the same buffers."""
        assert x.is_cuda and x.dtype == torch.float32 and x.is_contiguous()
        assert labels.is_cuda and labels.dtype == torch.int32 and labels.numel() == x.shape[0]
        self._n = x.shape[0]
        self._x, self._y = x, labels  # keep alive: the graph holds their addresses
        check(lib.vt_model_graph_capture(self._h, C.c_void_p(x.data_ptr()), C.c_void_p(labels.data_ptr()), self._n,
                                         self.lr, float(batch_size)), "vt_model_graph_capture")

    def capture_forward(self, x):
        """Captures the forward pass on the fixed device buffer x as a HIP graph (current
        training / eval mode); replay with step_graph(), results in self.probs_view(n)."""
        assert x.is_cuda and x.dtype == torch.float32 and x.is_contiguous()
        self._n = x.shape[0]
        self._x = x
        check(lib.vt_model_graph_capture_forward(self._h, C.c_void_p(x.data_ptr()), self._n, C.c_void_p(self._probs.data_ptr())),
              "vt_model_graph_capture_forward")
        return self._probs[:self._n]

    def step_graph(self):
        check(lib.vt_model_graph_launch(self._h, _stream_handle(self.device)), "vt_model_graph_launch")

    def metrics(self, labels=None):
        """(sum of cross-entropy terms, argmax labels) of the last forward, computed on the device."""
        loss = torch.zeros(1, dtype=torch.float32, device=self.device)
        am = torch.zeros(self._n, dtype=torch.int32, device=self.device)
        lp = C.c_void_p(labels.data_ptr()) if labels is not None else None
        check(lib.vt_model_metrics(self._h, lp, C.c_void_p(loss.data_ptr()), C.c_void_p(am.data_ptr()),
                                   _stream_handle(self.device)), "vt_model_metrics")
        return loss, am

    def _sync_for_host_io(self):
        # the library's checkpoint I/O copies with the null stream: finish the caller's stream first
        torch.cuda.current_stream(self.device).synchronize()

    def save(self, path):
        self._sync_for_host_io()
        check(lib.vt_model_save(self._h, str(path).encode()), "vt_model_save(%s)" % path)

    def load(self, path):
        self._sync_for_host_io()
        check(lib.vt_model_load(self._h, str(path).encode()), "vt_model_load(%s)" % path)

    def save_v2(self, path, epoch=0, step=0, loader_seed=42, loader_resets=0):
        """Reference v1 bytes + trailer with the transformer block, BatchNorm1D statistics and the
        training position (still loadable by the reference, which never reads past the fc layer)."""
        from ._lib import VtTrainState
        st = VtTrainState(2, int(epoch), int(step), int(loader_seed), int(loader_resets))
        self._sync_for_host_io()
        check(lib.vt_model_save_v2(self._h, C.byref(st), str(path).encode()), "vt_model_save_v2(%s)" % path)

    def load_v2(self, path):
        """Loads a v1 or v2 file; returns dict(version, epoch, step, loader_seed, loader_resets)."""
        from ._lib import VtTrainState
        st = VtTrainState()
        self._sync_for_host_io()
        check(lib.vt_model_load_v2(self._h, C.byref(st), str(path).encode()), "vt_model_load_v2(%s)" % path)
        return {"version": st.version, "epoch": st.epoch, "step": st.step, "loader_seed": st.loader_seed,
                "loader_resets": st.loader_resets}

    # ---- tooling -------------------------------------------------------------------------
    def profile(self, launcher=None):
        """Time one launcher ("projector_bwd"), every launcher ("*") or none (None) with HIP events."""
        check(lib.vt_model_profile(self._h, launcher.encode() if launcher else None))

    def profile_read(self):
        """{launcher: (total_ms, launches)} since the last profile() call (synchronises)."""
        out = {}
        n = lib.vt_model_profile_read(self._h, -1, None, 0, None, None)
        for k in range(max(n, 0)):
            name = C.create_string_buffer(64)
            ms, cnt = C.c_float(), C.c_int()
            lib.vt_model_profile_read(self._h, k, name, 64, C.byref(ms), C.byref(cnt))
            if cnt.value:
                out[name.value.decode()] = (ms.value, cnt.value)
        return out

    def profile_timeline(self):
        """[(launcher, start_us, end_us)] of every launch timed since profile(), relative to the first."""
        out = []
        n = lib.vt_model_profile_timeline(self._h, -1, None, 0, None, None)
        for k in range(max(n, 0)):
            name = C.create_string_buffer(64)
            a, b = C.c_float(), C.c_float()
            lib.vt_model_profile_timeline(self._h, k, name, 64, C.byref(a), C.byref(b))
            out.append((name.value.decode(), a.value, b.value))
        return out

    def read(self, name):
        n = lib.vt_model_read(self._h, name.encode(), None, 0)
        if n < 0:
            raise KeyError(name)
        out = np.zeros(n, dtype=np.float32)
        got = lib.vt_model_read(self._h, name.encode(), C.c_void_p(out.ctypes.data), n)
        if got != n:
            raise RuntimeError("vt_model_read(%s) -> %d" % (name, got))
        return out

    def set_deferred_attention(self, on):
        """Run the attention block's backward and weight update beside the next step's forward
        (vt_model_set_deferred_attention); same bytes.  Readers of the arenas go through join_attention()."""
        check(lib.vt_model_set_deferred_attention(self._h, int(bool(on))), "vt_model_set_deferred_attention")

    def join_attention(self):
        """Makes the current stream wait for a deferred attention backward / update still in flight; call
        before touching self.grads / self.arena on the device (collectives, optimizers of the caller)."""
        check(lib.vt_model_attention_join(self._h, _stream_handle(self.device)), "vt_model_attention_join")

    def _settled(self):
        # arena readers and writers below run on the current stream of the model's device: order them behind
        # work the model still has in flight on its own streams
        self.join_attention()
        return torch.cuda.stream(torch.cuda.current_stream(self.device))

    def get_arena(self):
        with self._settled():
            return self.arena.detach().cpu().numpy()

    def set_arena(self, host):
        host = np.ascontiguousarray(host, dtype=np.float32)
        assert host.size == self.layout.n_total
        with self._settled():
            self.arena.copy_(torch.from_numpy(host))

    def get_grads(self):
        with self._settled():
            return self.grads.detach().cpu().numpy()


# ---------------------------------------------------------------------------------------------
# Data path (reference utils/load_dataset.hpp)
# ---------------------------------------------------------------------------------------------
class SelfAttention:
    """Host mirror of the reference's SelfAttention layer (layers/self_attention.hpp): forward(x),
    backward(delta), get_input_deltas(), update_weights(batch_size), zero_grad().  Weights are the two
    bias-free DenseLayers' [channels, attn_dim] matrices, Xavier-uniform (dense_layer.hpp:22-23); a fresh
    layer draws them with numpy, so reference weights come in through set_parameters().  Like the
    reference layer, update_weights does nothing unless an optimizer was attached to the layer itself
    (self_attention.hpp:138-145): attached_optimizer=True stands for that."""

    def __init__(self, channels, attn_dim, rng_seed=42, attached_optimizer=False, lr=0.001):
        self.D, self.Dk, self.lr, self.attached = int(channels), int(attn_dim), float(lr), bool(attached_optimizer)
        limit = np.float32(math.sqrt(6.0 / (channels + attn_dim)))
        g = np.random.default_rng(rng_seed)
        self.wq = torch.from_numpy(g.uniform(-limit, limit, size=(channels, attn_dim)).astype(np.float32)).cuda()
        self.wk = torch.from_numpy(g.uniform(-limit, limit, size=(channels, attn_dim)).astype(np.float32)).cuda()
        self.g_wq, self.g_wk = torch.zeros_like(self.wq), torch.zeros_like(self.wk)
        self._x = None

    def set_parameters(self, key_w, query_w):
        self.wk = torch.as_tensor(key_w, dtype=torch.float32).contiguous().cuda()
        self.wq = torch.as_tensor(query_w, dtype=torch.float32).contiguous().cuda()
        assert tuple(self.wk.shape) == (self.D, self.Dk) and tuple(self.wq.shape) == (self.D, self.Dk)

    def forward(self, x):
        assert x.is_cuda and x.dtype == torch.float32 and x.is_contiguous() and x.dim() == 3 and x.shape[2] == self.D
        B, T, D = x.shape
        dev = x.device
        self._x = x
        self._key = torch.empty((B, T, self.Dk), device=dev)
        self._query = torch.empty((B, T, self.Dk), device=dev)
        self._attn = torch.empty((B, T, T), device=dev)
        out = torch.empty((B, T, D), device=dev)
        pv = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731
        check(lib.vt_self_attention_fwd(pv(x), pv(self.wk), pv(self.wq), pv(self._key), pv(self._query), pv(self._attn),
                                        pv(out), B, T, D, self.Dk, _stream_handle()), "vt_self_attention_fwd")
        return out

    def backward(self, delta):
        if self._x is None:
            raise RuntimeError("SelfAttention: No hay fuente de delta para backward.")
        x = self._x
        B, T, D = x.shape
        assert delta.is_cuda and delta.dtype == torch.float32 and delta.is_contiguous() and delta.shape == x.shape
        dev = x.device
        ds = torch.empty((B, T, T), device=dev)
        dkey = torch.empty((B, T, self.Dk), device=dev)
        dquery = torch.empty((B, T, self.Dk), device=dev)
        self._dx = torch.empty((B, T, D), device=dev)
        ws = torch.empty(1 << 20, device=dev)
        pv = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731
        check(lib.vt_self_attention_bwd(pv(x), pv(self.wk), pv(self.wq), pv(self._key), pv(self._query), pv(self._attn),
                                        pv(delta), pv(ds), pv(dkey), pv(dquery), pv(self.g_wk), pv(self.g_wq), pv(self._dx),
                                        pv(ws), ws.numel(), B, T, D, self.Dk, _stream_handle()), "vt_self_attention_bwd")

    def get_input_deltas(self):
        return self._dx

    def update_weights(self, batch_size):
        if not self.attached:
            return
        for w, g in ((self.wq, self.g_wq), (self.wk, self.g_wk)):
            check(lib.vt_sgd_update(C.c_void_p(w.data_ptr()), C.c_void_p(g.data_ptr()), w.numel(), self.lr, float(batch_size),
                                    0, _stream_handle()), "vt_sgd_update")

    def zero_grad(self):
        self.g_wq.zero_()
        self.g_wk.zero_()


class Dataset:
    def __init__(self, images, labels, images_u8=None):
        self.images = images  # float32 [S, rows*cols] = u8 / 255.0f
        self.labels = labels  # int32 [S]
        self.images_u8 = images_u8  # the IDX pixel bytes [S, rows*cols], when loaded from a file


def load_dataset(image_path, label_path, max_samples=9999999):
    """load_dataset (load_dataset.hpp:87-96): IDX images (u8/255.0f) and labels."""
    n, r, c = C.c_int(), C.c_int(), C.c_int()
    check(lib.vt_idx_read_images(str(image_path).encode(), max_samples, None, C.byref(n), C.byref(r), C.byref(c)),
          "No se pudo abrir el archivo de imágenes")
    imgs = np.zeros((n.value, r.value * c.value), dtype=np.float32)
    check(lib.vt_idx_read_images(str(image_path).encode(), max_samples, C.c_void_p(imgs.ctypes.data), C.byref(n),
                                 C.byref(r), C.byref(c)))
    m = C.c_int()
    check(lib.vt_idx_read_labels(str(label_path).encode(), max_samples, None, C.byref(m)),
          "No se pudo abrir el archivo de etiquetas")
    lbl = np.zeros(m.value, dtype=np.int32)
    check(lib.vt_idx_read_labels(str(label_path).encode(), max_samples, C.c_void_p(lbl.ctypes.data), C.byref(m)))
    if imgs.shape[0] != lbl.shape[0]:
        raise RuntimeError("El número de imágenes y etiquetas no coincide")
    raw = np.fromfile(str(image_path), dtype=np.uint8, offset=16, count=imgs.size)
    u8 = np.zeros(imgs.size, dtype=np.uint8)  # a short file leaves the missing pixels at 0, as the float path does
    u8[:raw.size] = raw
    return Dataset(imgs, lbl, u8.reshape(imgs.shape))


def load_image_as_vector(path, target_w=28, target_h=28):
    """load_image_as_vector (utils/load_image_stb.hpp:9-26): one grey float per pixel of a PNG,
    pixel / 255.0f, the grey byte exactly as the reference's decoder forms it."""
    w, h = C.c_int(0), C.c_int(0)
    if lib.vt_image_read_gray(str(path).encode(), None, 0, C.byref(w), C.byref(h)) != 0:
        raise RuntimeError("No se pudo cargar la imagen: %s" % path)
    if w.value != target_w or h.value != target_h:
        raise RuntimeError("La imagen debe ser de tamaño 28x28")
    raw = np.zeros(w.value * h.value, dtype=np.uint8)
    if lib.vt_image_read_gray(str(path).encode(), C.c_void_p(raw.ctypes.data), raw.size, C.byref(w), C.byref(h)) != 0:
        raise RuntimeError("No se pudo cargar la imagen: %s" % path)
    return raw.astype(np.float32) / np.float32(255.0)


def load_custom_images_from_folder(folder_path):
    """load_custom_images_from_folder (utils/load_dataset.hpp:167-209): .png/.jpg/.jpeg files in
    alphabetical order of their names; files that do not load are reported and skipped."""
    import sys
    names = sorted(f for f in os.listdir(folder_path)
                   if os.path.isfile(os.path.join(folder_path, f)) and os.path.splitext(f)[1] in (".png", ".jpg", ".jpeg"))
    images = []
    for f in names:
        try:
            images.append(load_image_as_vector(os.path.join(folder_path, f)))
        except RuntimeError as e:
            print("Error al cargar %s: %s" % (os.path.join(folder_path, f), e), file=sys.stderr)
    arr = np.stack(images) if images else np.zeros((0, 784), dtype=np.float32)
    return Dataset(arr, np.zeros(len(images), dtype=np.int32), None)


def loader_indices(n, seed=42, resets=0):
    out = np.zeros(n, dtype=np.int32)
    check(lib.vt_loader_indices(n, seed, resets, C.c_void_p(out.ctypes.data)))
    return out


class DataLoader:
    """DataLoader (load_dataset.hpp:99-165): shuffles in the constructor and at every reset()
    with one mt19937(seed); the last batch may be short.  The dataset is kept resident on the
    device; a batch is an index gather on the GPU."""

    def __init__(self, images, labels, batch_size, channels=1, height=28, width=28, num_classes=10, seed=42,
                 device=None):
        if len(images) != len(labels):
            raise RuntimeError("Mismatch entre imágenes y etiquetas")
        self.n = len(images)
        self.batch_size = int(batch_size)
        self.shape = (channels, height, width)
        self.seed = seed
        self._resets = 0
        self.index = 0
        self.indices = loader_indices(self.n, seed, 0)
        dev = device or torch.device("cuda", torch.cuda.current_device())
        # uint8 input (Dataset.images_u8): the dataset stays resident as bytes and a batch is one
        # gather launch with the /255.0f fused in; float input: resident floats, same gather in torch
        self._u8 = np.asarray(images).dtype == np.uint8
        if self._u8:
            self._images = torch.from_numpy(np.ascontiguousarray(images, dtype=np.uint8)).to(dev)
        else:
            self._images = torch.from_numpy(np.ascontiguousarray(images, dtype=np.float32)).to(dev)
        self._labels = torch.from_numpy(np.ascontiguousarray(labels, dtype=np.int32)).to(dev)
        self._idx_dev = torch.from_numpy(self.indices.astype(np.int32 if self._u8 else np.int64)).to(dev)

    def reset(self):
        self.index = 0
        self._resets += 1
        self.indices = loader_indices(self.n, self.seed, self._resets)
        self._idx_dev = torch.from_numpy(self.indices.astype(np.int32 if self._u8 else np.int64)).to(self._images.device)

    def has_next(self):
        return self.index < self.n

    def total_batches(self):
        return (self.n + self.batch_size - 1) // self.batch_size

    def skip(self, count_images):
        """Advances past the next `count_images` images without gathering them (another rank evaluates them)."""
        self.index = min(self.index + int(count_images), self.n)

    def next_batches(self, count):
        """`count` consecutive batches as one tensor pair (for grouped evaluation sweeps)."""
        saved = self.batch_size
        self.batch_size = saved * int(count)
        try:
            return self.next_batch()
        finally:
            self.batch_size = saved

    def next_batch(self):
        end = min(self.index + self.batch_size, self.n)
        sel = self._idx_dev[self.index:end]
        nb = end - self.index
        self.index = end
        if self._u8:
            pixels = self._images.shape[1]
            x = torch.empty((nb,) + self.shape, dtype=torch.float32, device=self._images.device)
            y = torch.empty(nb, dtype=torch.int32, device=self._images.device)
            check(lib.vt_loader_gather(C.c_void_p(self._images.data_ptr()), C.c_void_p(self._labels.data_ptr()),
                                       C.c_void_p(sel.data_ptr()), nb, pixels, C.c_void_p(x.data_ptr()),
                                       C.c_void_p(y.data_ptr()), _stream_handle()), "vt_loader_gather")
            return x, y
        x = self._images.index_select(0, sel).reshape(-1, *self.shape).contiguous()
        y = self._labels.index_select(0, sel).contiguous()
        return x, y


def class_stats(cm):
    """Tensor::precision_recall / f1_score (utils/tensor/tensor_cpu.hpp; printed by Trainer::evaluate_test,
    utils/trainer.hpp:131-153) from a confusion matrix cm[truth][pred], in single precision as there: per-class
    precision tp/column, recall tp/row (0 where the denominator is 0), F1 = 2pr/(p+r); macro means over the classes
    that occur among the true labels; weighted F1 by the number of true samples of each class."""
    cm = np.asarray(cm, dtype=np.int64)
    k = cm.shape[0]
    f = np.float32
    precision, recall, f1 = np.zeros(k, f), np.zeros(k, f), np.zeros(k, f)
    mp = mr = mf = wf = f(0)
    present = total = 0
    for c in range(k):
        col, row, tp = int(cm[:, c].sum()), int(cm[c, :].sum()), f(cm[c, c])
        if col > 0:
            precision[c] = tp / f(col)
        if row > 0:
            recall[c] = tp / f(row)
        if precision[c] + recall[c] > 0:
            f1[c] = f(2) * (precision[c] * recall[c]) / (precision[c] + recall[c])
        if row > 0:
            mp, mr, mf = mp + precision[c], mr + recall[c], mf + f1[c]
            wf = wf + f1[c] * f(row)
            total += row
            present += 1
    if present > 0:
        mp, mr, mf = mp / f(present), mr / f(present), mf / f(present)
    if total > 0:
        wf = wf / f(total)
    return {"precision": precision, "recall": recall, "f1": f1, "macro_precision": float(mp), "macro_recall": float(mr),
            "macro_f1": float(mf), "weighted_f1": float(wf)}


class Trainer:
    """Trainer (utils/trainer.hpp:9-243): train / evaluate / evaluate_test / predict.

    world / rank (one process per GPU, torch.distributed initialised by the caller): evaluation sweeps are sharded
    "replicas only": the chunks of consecutive loader batches are dealt round-robin to the ranks, every loader batch
    keeps its own BatchNorm1D statistics, and the per-batch loss sums, hit counts, predicted labels and the confusion
    matrix are merged afterwards with one sum over the ranks (each entry is written by exactly one rank); every rank
    then holds, and rank 0 prints, what the single-process sweep computes."""

    def __init__(self, model, train_loader, val_loader, num_classes, batch_size, test_loader=None, out=print, world=1, rank=0,
                 process_group=None):
        self.model, self.train_loader, self.val_loader, self.test_loader = model, train_loader, val_loader, test_loader
        self.num_classes, self.batch_size, self.out = num_classes, batch_size, out
        self.world, self.rank, self.group = int(world), int(rank), process_group
        if self.world > 1 and self.rank != 0:
            self.out = lambda *a, **k: None  # rank 0 prints

    def _merge_over_ranks(self, tensors):
        """In-place sum over the ranks of small device tensors (on the device over RCCL, through the host otherwise)."""
        import torch.distributed as dist
        on_device = dist.get_backend(self.group) == "nccl"
        for t in tensors:
            if on_device:
                dist.all_reduce(t, group=self.group)
            else:
                h = t.cpu()
                dist.all_reduce(h, group=self.group)
                t.copy_(h)

    def _epoch_eval(self, loader):
        """Forward-only sweep in eval mode; per-batch loss sums, hit counts and the confusion matrix
        stay on the device (vt_eval_accumulate) and are read back once at the end."""
        dev = self.model.device
        nb = loader.total_batches()
        k = self.num_classes
        loss_sums = torch.zeros(nb, dtype=torch.float32, device=dev)
        correct = torch.zeros(nb, dtype=torch.int32, device=dev)
        cm = torch.zeros(k * k, dtype=torch.int32, device=dev)
        preds, truth, sizes = [], [], []
        sharded = self.world > 1
        all_pred = torch.zeros(loader.n if sharded else 0, dtype=torch.int32, device=dev)  # merged by a sum: one writer per entry
        chunk = 0
        if sharded:
            # data-parallel ranks keep BatchNorm running statistics of their own; the model that is evaluated (and
            # saved) is rank 0's, so its frozen and state tensors go to every rank first (trainable weights already agree)
            import torch.distributed as dist
            tail = self.model.arena[self.model.layout.n_trainable:]
            if dist.get_backend(self.group) == "nccl":
                dist.broadcast(tail, src=0, group=self.group)
            else:
                h = tail.cpu()
                dist.broadcast(h, src=0, group=self.group)
                tail.copy_(h)
        self.model.set_training(False)
        b = 0
        bs = loader.batch_size
        # the grouped launch set exists for the fused headline kernels only; every other shape goes batch by batch
        grouped = bs % 4 == 0 and hasattr(loader, "next_batches") and getattr(self.model, "fused", False)
        per_call = (self.model.max_batch // bs) if grouped else 0
        if per_call >= 2:
            # whole chunks of loader batches per launch set (vt_model_forward_grouped): each batch
            # keeps its own BatchNorm1D statistics, as in the reference's batch-by-batch loop
            while loader.has_next():
                first_image = loader.index
                if sharded and chunk % self.world != self.rank:  # another rank's chunk
                    loader.skip(per_call * bs)
                    left = loader.index - first_image
                    while left > 0:
                        sizes.append(min(bs, left))
                        left -= bs
                        b += 1
                    chunk += 1
                    continue
                chunk += 1
                x, y = loader.next_batches(per_call)
                p = self.model.forward_grouped(x, bs)
                am = torch.empty(y.numel(), dtype=torch.int32, device=dev)
                check(lib.vt_eval_accumulate_grouped(C.c_void_p(p.data_ptr()), C.c_void_p(y.data_ptr()), y.numel(), k, bs, b,
                                                     C.c_void_p(loss_sums.data_ptr()), C.c_void_p(correct.data_ptr()),
                                                     C.c_void_p(cm.data_ptr()), C.c_void_p(am.data_ptr()),
                                                     _stream_handle()), "vt_eval_accumulate_grouped")
                preds.append(am)
                truth.append(y)
                if sharded:
                    all_pred[first_image:first_image + y.numel()] = am
                left = y.numel()
                while left > 0:
                    sizes.append(min(bs, left))
                    left -= bs
                    b += 1
        while loader.has_next():
            first_image = loader.index
            if sharded and chunk % self.world != self.rank:
                loader.skip(bs)
                sizes.append(loader.index - first_image)
                b += 1
                chunk += 1
                continue
            chunk += 1
            x, y = loader.next_batch()
            p = self.model.forward(x)
            am = torch.empty(y.numel(), dtype=torch.int32, device=dev)
            check(lib.vt_eval_accumulate(C.c_void_p(p.data_ptr()), C.c_void_p(y.data_ptr()), y.numel(), k, b,
                                         C.c_void_p(loss_sums.data_ptr()), C.c_void_p(correct.data_ptr()),
                                         C.c_void_p(cm.data_ptr()), C.c_void_p(am.data_ptr()), _stream_handle()),
                  "vt_eval_accumulate")
            preds.append(am)
            truth.append(y)
            if sharded:
                all_pred[first_image:first_image + y.numel()] = am
            sizes.append(y.numel())
            b += 1
        self.model.set_training(True)
        if sharded:
            self._merge_over_ranks([loss_sums, correct, cm, all_pred])
            order = torch.from_numpy(loader.indices.astype(np.int64)).to(dev)
            preds, truth = [all_pred], [loader._labels.index_select(0, order)]
        sizes = np.asarray(sizes, dtype=np.float64)
        ls = loss_sums[:b].cpu().numpy().astype(np.float64) / sizes  # per-batch means, then their mean
        ac = correct[:b].cpu().numpy().astype(np.float64) / sizes    # (utils/trainer.hpp:112-127)
        self._last_cm = cm.cpu().numpy().reshape(k, k).astype(np.int64)
        return float(ls.mean()), float(ac.mean()), torch.cat(preds).cpu().numpy(), torch.cat(truth).cpu().numpy()

    def train(self, epochs, log_every=100, device="hip"):
        for epoch in range(epochs):
            t0 = time.time()
            self.out("Epoch %d/%d" % (epoch + 1, epochs))
            self.train_loader.reset()
            losses, accs = [], []
            while self.train_loader.has_next():
                x, y = self.train_loader.next_batch()
                self.model.forward(x)
                self.model.backward(y)
                ls, am = self.model.metrics(y)
                self.model.update_weights(self.batch_size)
                losses.append(ls / y.numel())
                accs.append((am == y).float().mean())
            torch.cuda.synchronize()
            dt = time.time() - t0
            self.out("Train Loss: %.4f | Accuracy: %.4f%% | Time: %.4f s" % (
                float(torch.stack(losses).mean().item()), 100.0 * float(torch.stack(accs).mean().item()), dt))
            self.evaluate()
        if self.test_loader is not None:
            self.out("\n=== Evaluación final sobre conjunto de test ===")
            self.evaluate_test()

    def evaluate(self):
        self.val_loader.reset()
        loss, acc, _, _ = self._epoch_eval(self.val_loader)
        self.out("Val Loss: %.4f | Accuracy: %.4f%%" % (loss, acc * 100.0))
        return loss, acc

    def evaluate_test(self):
        if self.test_loader is None:
            self.out("[WARN] No se proporcionó test_loader")
            return None
        self.test_loader.reset()
        loss, acc, pred, truth = self._epoch_eval(self.test_loader)
        self.out("[TEST] Final Evaluation — Loss: %.4f | Accuracy: %.4f%%" % (loss, acc * 100.0))
        k = self.num_classes
        cm = self._last_cm  # accumulated on the device, confusion[truth][pred]
        st = class_stats(cm)
        self.out(" - F1 Score (macro):    %.4f" % st["macro_f1"])
        self.out(" - F1 Score (weighted): %.4f" % st["weighted_f1"])
        self.out(" - Precision (macro):   %.4f" % st["macro_precision"])
        self.out(" - Recall (macro):      %.4f" % st["macro_recall"])
        self.out("\n📊 Per-Class Metrics:")
        self.out("%8s%12s%12s%12s" % ("Class", "Precision", "Recall", "F1 Score"))
        self.out("-" * 44)
        for i_ in range(k):
            self.out("%8d%12.4f%12.4f%12.4f" % (i_, st["precision"][i_], st["recall"][i_], st["f1"][i_]))
        self.out("\n🧩 Matriz de Confusión:\n    " + "".join("%4d" % j for j in range(k)))
        for i_ in range(k):
            self.out("%2d: " % i_ + "".join("%4d" % cm[i_, j] for j in range(k)))
        return loss, acc, pred, truth, cm

    def predict(self, image_data):
        img = np.asarray(image_data, dtype=np.float32)
        if img.size != 28 * 28:
            raise RuntimeError("La imagen debe ser de tamaño 28x28")
        x = torch.from_numpy(img.reshape(1, 1, 28, 28)).to(self.model.device)
        self.model.set_training(False)
        p = self.model.forward(x)
        return int(torch.argmax(p[0]).item())
