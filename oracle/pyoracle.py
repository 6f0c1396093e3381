"""ctypes binding of the CPU oracle (oracle/vtcnn_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, by __graft_entry__.smoke() and by the
`cpu_baseline` leg of bench.py.  The product package (vit_amd/) never imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "libvtcnn_oracle.so")

VT_NAME_LEN = 40
VT_MAX_TENSORS = 96


class VtDims(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("image_channels", "input_dim", "tokens", "token_channels",
                                       "vt_channels", "heads", "num_classes")]


class VtTensorInfo(C.Structure):
    _fields_ = [("name", C.c_char * VT_NAME_LEN), ("offset", C.c_int), ("size", C.c_int), ("ndim", C.c_int),
                ("shape", C.c_int * 4), ("kind", C.c_int)]


class VtLayout(C.Structure):
    _fields_ = [("n_tensors", C.c_int), ("n_trainable", C.c_int), ("n_total", C.c_int),
                ("n_trainable_exact", C.c_int), ("t", VtTensorInfo * VT_MAX_TENSORS)]


def build():
    """Compile the oracle library (and, where the reference checkout exists, oracle/_ref)."""
    subprocess.check_call(["make", "-s", "-C", _HERE])


def _load(path=_LIB):
    if not os.path.exists(path):
        build()
    lib = C.CDLL(path)
    lib.vto_create.restype = C.c_void_p
    lib.vto_create.argtypes = [C.POINTER(VtDims), C.c_int]
    lib.vto_destroy.argtypes = [C.c_void_p]
    lib.vto_layout.argtypes = [C.POINTER(VtDims), C.POINTER(VtLayout)]
    lib.vto_forward.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
    lib.vto_backward.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.vto_loss.restype = C.c_double if path.endswith("64.so") else C.c_float
    lib.vto_loss.argtypes = [C.c_void_p, C.c_void_p]
    lib.vto_argmax.argtypes = [C.c_void_p, C.c_void_p]
    lib.vto_get.restype = C.POINTER(C.c_double if path.endswith("64.so") else C.c_float)
    lib.vto_get.argtypes = [C.c_void_p, C.c_char_p, C.POINTER(C.c_long)]
    real = C.c_double if path.endswith("64.so") else C.c_float
    lib.vto_sgd.argtypes = [C.POINTER(VtDims), C.c_void_p, C.c_void_p, real, real]
    return lib


_lib = None
_lib64 = None


def lib():
    global _lib
    if _lib is None:
        _lib = _load()
    return _lib


def lib64():
    global _lib64
    if _lib64 is None:
        _lib64 = _load(os.path.join(_HERE, "libvtcnn_oracle64.so"))
    return _lib64


def make_dims(num_classes=10, image_channels=1, input_dim=28, tokens=16, token_channels=32, vt_channels=16, heads=2):
    return VtDims(image_channels, input_dim, tokens, token_channels, vt_channels, heads, num_classes)


class Layout:
    """name -> (offset, size, shape, kind) view of the canonical arena layout."""

    def __init__(self, dims):
        lo = VtLayout()
        if lib().vto_layout(C.byref(dims), C.byref(lo)) != 0:
            raise ValueError("unsupported dims")
        self.n_trainable = lo.n_trainable
        self.n_total = lo.n_total
        self.n_trainable_exact = lo.n_trainable_exact
        self.tensors = {}
        self.order = []
        for i in range(lo.n_tensors):
            t = lo.t[i]
            name = t.name.decode()
            self.tensors[name] = (t.offset, t.size, tuple(t.shape[:t.ndim]), t.kind)
            self.order.append(name)

    def view(self, arena, name):
        off, size, shape, _ = self.tensors[name]
        return arena[off:off + size].reshape(shape)

    def pack(self, named, prefix="", kinds=(0, 1, 2)):
        """Builds an arena (or, for kinds=(0,), a gradient arena) from a dict of named arrays."""
        n = self.n_trainable if tuple(kinds) == (0,) else self.n_total
        arena = np.zeros(n, dtype=np.float32)
        for name in self.order:
            off, size, shape, kind = self.tensors[name]
            if kind not in kinds:
                continue
            arena[off:off + size] = np.asarray(named[prefix + name], dtype=np.float32).ravel()
        return arena


class Oracle:
    """One VTCNN step on the CPU.  f64=True runs the double-precision build (arena, inputs and
    outputs are float64 arrays then)."""

    def __init__(self, dims, n, f64=False):
        self.dims = dims
        self.n = n
        self.layout = Layout(dims)
        self.lib = lib64() if f64 else lib()
        self.dtype = np.float64 if f64 else np.float32
        self.ctx = self.lib.vto_create(C.byref(dims), n)
        if not self.ctx:
            raise ValueError("vto_create failed")

    def __del__(self):
        if getattr(self, "ctx", None):
            self.lib.vto_destroy(self.ctx)
            self.ctx = None

    def forward(self, arena, x, training=True):
        assert arena.dtype == self.dtype and arena.size == self.layout.n_total and arena.flags.c_contiguous
        x = np.ascontiguousarray(x, dtype=self.dtype)
        assert x.size == self.n * self.dims.image_channels * self.dims.input_dim ** 2
        self._x = x
        self.lib.vto_forward(self.ctx, arena.ctypes.data, x.ctypes.data, int(training))
        return self.get("fwd.preds").reshape(self.n, self.dims.num_classes)

    def loss(self, labels):
        labels = np.ascontiguousarray(labels, dtype=np.int32)
        return float(self.lib.vto_loss(self.ctx, labels.ctypes.data))

    def argmax(self):
        out = np.zeros(self.n, dtype=np.int32)
        self.lib.vto_argmax(self.ctx, out.ctypes.data)
        return out

    def backward(self, arena, labels):
        labels = np.ascontiguousarray(labels, dtype=np.int32)
        grads = np.zeros(self.layout.n_trainable, dtype=self.dtype)
        self.lib.vto_backward(self.ctx, arena.ctypes.data, self._x.ctypes.data, labels.ctypes.data, grads.ctypes.data)
        return grads

    def set_routing(self, pool_slot=None, relu_mask=None):
        """Imposes another evaluation's max-pool slots / ReLU mask (uint8 arrays), or clears them."""
        self._slot = None if pool_slot is None else np.ascontiguousarray(pool_slot, dtype=np.uint8)
        self._mask = None if relu_mask is None else np.ascontiguousarray(relu_mask, dtype=np.uint8)
        self.lib.vto_set_routing.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        self.lib.vto_set_routing(self.ctx, None if self._slot is None else self._slot.ctypes.data,
                                 None if self._mask is None else self._mask.ctypes.data)

    def pool_decisions(self):
        """(slot uint8 [N*Q], gap [N*Q]) of the oracle's own max-pool decisions in the last forward."""
        q = self.n * self.dims.vt_channels * 196
        slot = np.zeros(q, dtype=np.uint8)
        gap = np.zeros(q, dtype=self.dtype)
        self.lib.vto_pool_decisions.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        self.lib.vto_pool_decisions(self.ctx, slot.ctypes.data, gap.ctypes.data)
        return slot, gap

    def sgd(self, arena, grads, lr=0.001, divisor=32.0):
        self.lib.vto_sgd(C.byref(self.dims), arena.ctypes.data, grads.ctypes.data, lr, divisor)

    def get(self, name):
        cnt = C.c_long(0)
        p = self.lib.vto_get(self.ctx, name.encode(), C.byref(cnt))
        if not p:
            raise KeyError(name)
        return np.ctypeslib.as_array(p, shape=(cnt.value,)).copy()
