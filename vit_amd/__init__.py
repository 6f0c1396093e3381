"""vit_amd: MI355X-native (gfx950, HIP) engine for the VTCNN train / predict hot path.

The compute lives in libvtcnn_hip.so (hand-written HIP kernels behind the C ABI declared in
include/vtcnn_hip.h); this package is the thin host-side mirror of the reference's
Model / Trainer / DataLoader interface on top of it.  There is no CPU fallback: importing
`vit_amd.engine` without the built library raises.
"""
from . import _lib  # noqa: F401
from .engine import VTCNN, Dims, DataLoader, Trainer, SelfAttention, load_dataset  # noqa: F401

__all__ = ["VTCNN", "Dims", "DataLoader", "Trainer", "SelfAttention", "load_dataset"]
