"""Data-parallel training over the GPUs of one node: one process per GPU, torch.distributed
(backend "nccl" = RCCL over xGMI on ROCm; "gloo" in CPU tests).

The reference has no multi-device code at all; this is the extension the hot path needs to scale.
Semantics (the oracle is "the reference run on each rank's mini-batch from identical weights,
gradients summed on the host, one SGD step"):

  * the unit of work is one reference mini-batch (BatchNorm couples the samples of a batch, so a
    batch is never split): in step t rank r trains mini-batch t*world + r of the shuffled order;
  * no activation is exchanged; the only collective is ONE sum all-reduce per step of the flat
    gradient arena (39 786 floats = 159 KB at the reference's hyper-parameters: latency bound, so a
    single bucket is right; bucketing would only add launches);
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

    def __init__(self, model, process_group=None, world=None, rank=None):
        import torch.distributed as dist
        self.model = model
        self.dist = dist
        self.group = process_group
        self.world = world if world is not None else (dist.get_world_size(process_group) if dist.is_initialized() else 1)
        self.rank = rank if rank is not None else (dist.get_rank(process_group) if dist.is_initialized() else 0)

    def all_reduce_grads(self, async_op=False):
        """One flat sum all-reduce of the whole gradient arena."""
        if self.world == 1:
            return None
        return self.dist.all_reduce(self.model.grads, op=self.dist.ReduceOp.SUM, group=self.group, async_op=async_op)

    def train_step(self, x, y, batch_size):
        """forward + backward on this rank's mini-batch, gradient all-reduce, SGD with the
        reference's divisor (its constant Trainer::batch_size, utils/trainer.hpp:53) times world."""
        probs = self.model.forward(x)
        self.model.backward(y)
        self.all_reduce_grads()
        self.model.update_weights(float(batch_size) * self.world)
        return probs

    def weights_in_sync(self):
        """True when the trainable arena is bit-identical on every rank (debug / test aid)."""
        if self.world == 1:
            return True
        n = self.model.layout.n_trainable
        mine = self.model.arena[:n].clone()
        lo, hi = mine.clone(), mine.clone()
        self.dist.all_reduce(lo, op=self.dist.ReduceOp.MIN, group=self.group)
        self.dist.all_reduce(hi, op=self.dist.ReduceOp.MAX, group=self.group)
        return bool(torch.equal(lo, hi))
