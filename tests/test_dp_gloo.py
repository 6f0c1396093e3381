"""Data-parallel path (vit_amd/parallel.py) on CPU: world_size 2, gloo backend.

There is no GPU here, so the compute backend behind the DataParallel wrapper is the CPU oracle
(test infrastructure); what is under test is the DP logic itself: batch sharding, the single flat
gradient all-reduce, the world*batch divisor and weight synchronisation.  Expected values come from
the REAL reference: tests/golden/dp_g2_n4_k8.npz holds, for two mini-batches of 4 images, the
gradients the reference computes from identical seed-42 weights (oracle/ref_harness.cpp `dp`)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from conftest import load_golden


class OracleModel:
    """Minimal model facade (forward / backward / update_weights / grads / arena / layout)."""

    def __init__(self, k, n, arena):
        from oracle.pyoracle import Oracle, make_dims
        self.o = Oracle(make_dims(num_classes=k), n)
        self.layout = self.o.layout
        self.arena_np = arena.copy()
        self.arena = torch.from_numpy(self.arena_np)  # shares memory
        self.grads = torch.zeros(self.layout.n_trainable, dtype=torch.float32)
        self.lr = 0.001

    def forward(self, x):
        return torch.from_numpy(self.o.forward(self.arena_np, x.numpy(), training=True))

    def backward(self, y):
        self.grads.copy_(torch.from_numpy(self.o.backward(self.arena_np, y.numpy())))

    def update_weights(self, divisor):
        self.o.sgd(self.arena_np, self.grads.numpy(), self.lr, divisor)
        self.grads.zero_()


class PhasedOracleModel(OracleModel):
    """Adds the two-call backward of the device model.  The oracle has one backward only, so each
    call runs it whole and keeps its own part: the first everything outside the attention block,
    the second (with the attention weights as they are THEN) the block itself."""

    def attention_range(self):
        a = self.layout.tensors["mha.q0.W"][0]
        b = self.layout.tensors["mha.o.W"][0]
        return a, b - a

    def backward_main(self, y):
        self._y = y
        a, c = self.attention_range()
        g = torch.from_numpy(self.o.backward(self.arena_np, y.numpy()))
        self.grads[:a].copy_(g[:a])
        self.grads[a + c:].copy_(g[a + c:])

    def backward_attention(self):
        a, c = self.attention_range()
        g = torch.from_numpy(self.o.backward(self.arena_np, self._y.numpy()))
        self.grads[a:a + c].copy_(g[a:a + c])

    def update_weights_range(self, divisor, first, count):
        g = self.grads.numpy()[first:first + count]
        w = self.arena_np[first:first + count]
        w -= (g * np.float32(1.0 / divisor)) * np.float32(self.lr)  # two roundings, as SGD::update
        g[:] = 0


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_dir):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle.pyoracle import Layout, make_dims
    from vit_amd.parallel import DataParallel
    g = load_golden("dp_g2_n4_k8.npz")
    init = load_golden("step_n2_k8_full.npz")
    lo = Layout(make_dims(num_classes=8))
    arena0 = lo.pack(init, "p0.")
    m = OracleModel(8, 4, arena0)
    dp = DataParallel(m)
    assert dp.world == world and dp.rank == rank
    x = torch.from_numpy(g["in%d.X" % rank])
    y = torch.from_numpy(g["in%d.labels" % rank].astype(np.int32))
    # step by hand to observe the all-reduced gradients before the update clears them
    m.forward(x)
    m.backward(y)
    dp.all_reduce_grads()
    summed = m.grads.numpy().copy()
    m.update_weights(32.0 * world)
    ok_sync = dp.weights_in_sync()
    np.savez(os.path.join(out_dir, "rank%d.npz" % rank), summed=summed, arena=m.arena_np, sync=np.array(ok_sync))
    # and once more through train_step on swapped shards: still in sync afterwards
    x2 = torch.from_numpy(g["in%d.X" % (1 - rank)])
    y2 = torch.from_numpy(g["in%d.labels" % (1 - rank)].astype(np.int32))
    dp.train_step(x2, y2, 32)
    assert dp.weights_in_sync()
    # overlapped exchange (attention block trailing into the next step) against the plain one,
    # three steps from the same start: bit-identical weights on every rank
    plain, over = OracleModel(8, 4, arena0), PhasedOracleModel(8, 4, arena0)
    dp_plain, dp_over = DataParallel(plain), DataParallel(over, overlap=True)
    assert dp_over.overlap and not dp_plain.overlap
    a, c = over.attention_range()
    for t in range(3):
        xt, yt = (x, y) if t % 2 == 0 else (x2, y2)
        dp_plain.train_step(xt, yt, 32)
        dp_over.train_step(xt, yt, 32)
        nt = lo.n_trainable
        lag = np.flatnonzero(plain.arena_np[:nt] != over.arena_np[:nt])
        assert lag.size > 0 and lag.min() >= a and lag.max() < a + c  # only the attention block trails
    dp_over.finish()
    assert np.array_equal(plain.arena_np[:lo.n_trainable], over.arena_np[:lo.n_trainable])
    assert not over.grads.numpy().any()
    assert dp_over.weights_in_sync()
    dist.barrier()
    dist.destroy_process_group()


def test_dp_world2_matches_reference_gradient_sum(tmp_path):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    from oracle.pyoracle import Layout, make_dims
    g = load_golden("dp_g2_n4_k8.npz")
    init = load_golden("step_n2_k8_full.npz")
    lo = Layout(make_dims(num_classes=8))
    arena0 = lo.pack(init, "p0.")
    r = [dict(np.load(os.path.join(str(tmp_path), "rank%d.npz" % k))) for k in range(world)]
    assert bool(r[0]["sync"]) and bool(r[1]["sync"])
    assert np.array_equal(r[0]["summed"], r[1]["summed"])
    nt = lo.n_trainable  # running statistics are per rank by design; trainable weights must agree
    assert np.array_equal(r[0]["arena"][:nt], r[1]["arena"][:nt])
    assert not np.array_equal(r[0]["arena"][nt:], r[1]["arena"][nt:])
    for name, (off, size, shape, kind) in lo.tensors.items():
        if kind != 0:
            continue
        ref_sum = (g["g0." + name].astype(np.float32) + g["g1." + name].astype(np.float32)).ravel()
        assert np.array_equal(r[0]["summed"][off:off + size], ref_sum), name
        # w <- w - (g * 1/(world*32)) * lr, two roundings, as SGD::update does
        w0 = arena0[off:off + size]
        expect = w0 - (ref_sum * np.float32(1.0 / 64.0)) * np.float32(0.001)
        assert np.array_equal(r[0]["arena"][off:off + size], expect.astype(np.float32)), name
    # frozen tokenizer untouched
    for nm in ("tok.linear1.W", "tok.linear2.W"):
        assert np.array_equal(lo.view(r[0]["arena"], nm), lo.view(arena0, nm))


def test_shard_batches_partition():
    from vit_amd.parallel import shard_batches
    for total, world in ((374, 8), (10, 4), (7, 2), (3, 4), (16, 1)):
        shards = [shard_batches(total, world, r) for r in range(world)]
        flat = sorted(i for s in shards for i in s)
        assert flat == list(range((total // world) * world))
        assert len({len(s) for s in shards}) == 1
        for t in range(total // world):
            assert [s[t] for s in shards] == [t * world + r for r in range(world)]
