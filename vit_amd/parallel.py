"""Data-parallel training over the GPUs of one node: one process per GPU, torch.distributed
(backend "nccl" = RCCL over xGMI on ROCm; "gloo" in CPU tests).

The reference has no multi-device code at all; this is the extension the hot path needs to scale.
Semantics (the oracle is "the reference run on each rank's mini-batch from identical weights,
gradients summed on the host, one SGD step"):

  * the unit of work is one reference mini-batch (BatchNorm couples the samples of a batch, so a
    batch is never split): in step t rank r trains mini-batch t*world + r of the shuffled order;
  * no activation is exchanged; the only collective is a sum all-reduce of the flat gradient arena
    (39 786 floats = 159 KB at the reference's hyper-parameters: latency bound, not bandwidth bound);
  * by default the exchange is one blocking all-reduce between backward and update.  overlap=True
    (models with the two-call backward: backward_main / backward_attention) hides it instead:
    everything but the attention block's gradients is final after backward_main and is all-reduced
    while the attention backward (the longest kernels of the step) runs; the attention block's own
    gradients are all-reduced while the NEXT step's forward and backward_main run, which is legal
    because the forward never reads the attention weights (their output is discarded by the zero
    concat, layers/multi_head_attention.hpp:100) -- the update lands before the next
    backward_attention, the first reader.  The weights every kernel reads are the ones synchronous
    SGD would give it, bit for bit; finish() applies the trailing update before anything else
    (checkpoint, end of training) looks.  It is off by default because of what the hand-offs between
    the compute stream and the collective's stream cost on this hardware: rehearsed on one MI355X
    with a one-rank RCCL group (tools/probe_overlap.py, batch 1024) the blocking all-reduce adds
    8 us to a 363 us step, each collective issued mid-step 25-30 us; hiding pays only where the
    all-reduce itself is slower than that (more ranks, slower links);
  * every rank then applies  w <- w - lr * (sum_r g_r) * 1/(world * batch)  itself: weights stay
    bit-identical across ranks without a broadcast (identical seed-42 init, identical update);
  * BatchNorm running statistics are NOT exchanged: each rank keeps the statistics of its own
    batches, and rank 0's are the ones a checkpoint saves (the reference has no rule for this);
  * the frozen tokenizer stays frozen on every rank.

The model only has to offer forward / backward / update_weights and a flat `grads` tensor, so the
CPU tests can drive the same code with the oracle as the compute backend.
"""
import torch


def shard_batches(total_batches, world, rank):
    """Indices of the mini-batches rank `rank` trains: t*world + rank while a full round of
    `world` batches remains (a ragged last round is dropped so that every rank takes the same
    number of steps and the all-reduce never waits for a missing peer)."""
    rounds = total_batches // world
    return [t * world + rank for t in range(rounds)]


class DataParallel:
    """Wraps a model for synchronous data-parallel SGD."""

    def __init__(self, model, process_group=None, world=None, rank=None, overlap=None):
        import torch.distributed as dist
        self.model = model
        self.dist = dist
        self.group = process_group
        self.world = world if world is not None else (dist.get_world_size(process_group) if dist.is_initialized() else 1)
        self.rank = rank if rank is not None else (dist.get_rank(process_group) if dist.is_initialized() else 0)
        can = all(hasattr(model, a) for a in ("backward_main", "backward_attention", "attention_range", "update_weights_range"))
        if overlap and not can:
            raise ValueError("overlap=True needs a model with backward_main / backward_attention")
        self.overlap = bool(overlap)
        self._trailing = None  # (work handle or None, divisor): the attention block's exchange still in flight
        # a world of one has nothing to exchange; `always_exchange` still issues the collectives
        # (tests: the real backend's stream ordering on a single GPU)
        self.always_exchange = False

    def _sum(self, t):
        if self.world == 1 and not self.always_exchange:
            return None
        return self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM, group=self.group, async_op=True)

    def all_reduce_grads(self, async_op=False):
        """One flat sum all-reduce of the whole gradient arena."""
        if self.world == 1:
            return None
        if hasattr(self.model, "join_attention"):
            self.model.join_attention()  # a deferred attention backward must have written its gradients
        return self.dist.all_reduce(self.model.grads, op=self.dist.ReduceOp.SUM, group=self.group, async_op=async_op)

    def train_step(self, x, y, batch_size):
        """forward + backward on this rank's mini-batch, gradient all-reduce, SGD with the
        reference's divisor (its constant Trainer::batch_size, utils/trainer.hpp:53) times world."""
        m = self.model
        divisor = float(batch_size) * self.world
        if self.world == 1 and not self.overlap and not getattr(self, "always_exchange", False) and hasattr(m, "train_step"):
            return m.train_step(x, y, divisor)  # nothing to exchange: the update rides in the step's last launch
        if not self.overlap:
            probs = m.forward(x)
            m.backward(y)
            self.all_reduce_grads()
            m.update_weights(divisor)
            return probs
        first, count = m.attention_range()
        n = m.grads.numel()
        probs = m.forward(x)
        m.backward_main(y)
        # every gradient outside the attention block is final: exchange it under the attention backward
        # (a collective waits for the work already queued on the current stream, not for later work)
        head = [self._sum(m.grads[:first]), self._sum(m.grads[first + count:])]
        self.finish()  # last step's attention update, before its first reader
        m.backward_attention()
        for w in head:
            if w is not None:
                w.wait()  # the current stream waits; the host does not (nccl)
        m.update_weights_range(divisor, 0, first)
        m.update_weights_range(divisor, first + count, n - first - count)
        self._trailing = (self._sum(m.grads[first:first + count]), divisor)
        return probs

    def finish(self):
        """Applies the attention block's update still in flight from the last train_step (no-op
        otherwise).  Call before saving, evaluating attention weights or ending the run."""
        if self._trailing is None:
            return
        work, divisor = self._trailing
        self._trailing = None
        if work is not None:
            work.wait()
        first, count = self.model.attention_range()
        self.model.update_weights_range(divisor, first, count)

    def weights_in_sync(self):
        """True when the trainable arena is bit-identical on every rank (debug / test aid)."""
        self.finish()
        if self.world == 1:
            return True
        n = self.model.layout.n_trainable
        mine = self.model.arena[:n].clone()
        lo, hi = mine.clone(), mine.clone()
        self.dist.all_reduce(lo, op=self.dist.ReduceOp.MIN, group=self.group)
        self.dist.all_reduce(hi, op=self.dist.ReduceOp.MAX, group=self.group)
        return bool(torch.equal(lo, hi))
