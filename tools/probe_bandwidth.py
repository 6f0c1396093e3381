"""Device copy bandwidth with the access pattern of the per-image kernels (vt_probe_copy): one
contiguous region per workgroup against round-robin 4 KB chunks, and against nontemporal accesses, for a
few region sizes.

    PYTHONPATH=. python tools/probe_bandwidth.py
"""
import ctypes as C

import torch

from vit_amd._lib import check, lib


def rate(blocks, floats, mode, reps=20):
    src = torch.rand(blocks * floats, device="cuda")
    dst = torch.empty_like(src)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    for _ in range(3):
        check(lib.vt_probe_copy(C.c_void_p(src.data_ptr()), C.c_void_p(dst.data_ptr()), blocks, floats, mode, st))
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        check(lib.vt_probe_copy(C.c_void_p(src.data_ptr()), C.c_void_p(dst.data_ptr()), blocks, floats, mode, st))
    b.record()
    torch.cuda.synchronize()
    assert torch.equal(src, dst)
    us = a.elapsed_time(b) * 1e3 / reps
    return 2.0 * blocks * floats * 4 / us / 1e3, us  # GB/s (read + write), us per launch


if __name__ == "__main__":
    for blocks in (1024, 2048, 4096):
        for kb in (25, 50, 100):
            floats = kb * 256
            r0, t0 = rate(blocks, floats, 0)
            r1, t1 = rate(blocks, floats, 1)
            print("blocks %5d x %3d KB: per-workgroup regions %6.0f GB/s (%.1f us)   round-robin chunks %6.0f GB/s (%.1f us)"
                  % (blocks, kb, r0, t0, r1, t1))
            print("   nontemporal load+store %6.0f GB/s, store only %6.0f GB/s" % (rate(blocks, floats, 2)[0], rate(blocks, floats, 3)[0]))
