"""Data-parallel sharding of the minibatch path, one process per GPU.

Mirrors the reference's only parallel construct, the pthread fan-out of
CRF_Minibatch_GradAccumulator (trainers/accumulators/CRF_Minibatch_GradAccumulator.cpp):
  * rank r == stream r views the contiguous utterance range [r*floor(U/N), (r+1)*floor(U/N)),
    the last rank takes the remainder (io/CRF_FeatureStreamManager.cpp:425-464);
  * per step rank r processes floor(mb/N) + (r < mb mod N) utterances (:229-241,257);
  * a rank whose view is exhausted is inactive; the epoch ends when all are (:248-250,312);
  * gradient = (sum over active ranks) / n_active, scalars are plain sums (:277-308).
The collective is torch.distributed all_reduce: backend "nccl" is RCCL over xGMI on the MI355X
node, "gloo" on CPU for the world_size>1 tests.  No other collective exists on this path.
"""
import torch
import torch.distributed as dist


def view_range(n_utts, world, rank):
    per = n_utts // world
    lo = rank * per
    hi = n_utts if rank == world - 1 else (rank + 1) * per
    return lo, hi


def minibatch_share(minibatch, world, rank):
    return minibatch // world + (1 if rank < minibatch % world else 0)


class RankCursor:
    """Position of one rank inside its utterance view across the steps of an epoch."""

    def __init__(self, n_utts, world, rank):
        self.lo, self.hi = view_range(n_utts, world, rank)
        self.world, self.rank = world, rank
        self.pos = self.lo

    def rewind(self):
        self.pos = self.lo

    @property
    def active(self):
        return self.pos < self.hi

    def next_step(self, minibatch):
        """utterance indices of this rank for the next step ([] when the view is exhausted)"""
        if not self.active:
            return range(0)
        n = min(minibatch_share(minibatch, self.world, self.rank), self.hi - self.pos)
        r = range(self.pos, self.pos + n)
        self.pos += n
        return r


class MinibatchReducer:
    """The reference's join / sum / average (CRF_Minibatch_GradAccumulator.cpp:277-312) for one process per
    GPU, as ONE collective per SGD step: the gradient and the scalars {numer, zx, n_utts, active} live in one
    device buffer of lambda_len + 4 doubles that is all-reduced (sum) in place, then grad /= n_active.

    `grad` (a view of the first lambda_len entries) is what the engine accumulates into
    (Engine.set_grad_buffer(reducer.grad.data_ptr())); `tail` holds the four scalars.  Nothing is allocated
    and no host value travels to the device per step: the caller flips `set_active` only when its view runs
    out.  Inactive ranks must contribute a zero gradient.  All ranks call reduce() every step (the
    reference joins every stream's thread every step)."""

    def __init__(self, lambda_len, device, group=None):
        self.buf = torch.zeros(lambda_len + 4, dtype=torch.float64, device=device)
        self.grad = self.buf[:lambda_len]
        self.tail = self.buf[lambda_len:]
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self._one = torch.ones((), dtype=torch.float64, device=device)
        self._active = torch.ones((), dtype=torch.float64, device=device)
        self._nact = torch.ones((), dtype=torch.float64, device=device)

    def set_active(self, active):
        self._active.fill_(1.0 if active else 0.0)

    def reduce(self):
        """in place; returns n_active as a 0-d device tensor (no host synchronisation)"""
        self.tail[3].copy_(self._active)
        if self.world > 1:
            dist.all_reduce(self.buf, op=dist.ReduceOp.SUM, group=self.group)
        torch.maximum(self.tail[3], self._one, out=self._nact)
        self.grad.div_(self._nact)
        return self.tail[3]


def reduce_minibatch(grad, scalars, active, group=None):
    """In place: grad <- sum_ranks(grad) / n_active ; scalars = [numer, zx, n_utts] summed.
    Returns n_active as a 0-d tensor on grad's device.  Convenience form for tests and small tools (it
    allocates and runs two collectives); training loops use MinibatchReducer."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    extra = torch.zeros(scalars.numel() + 1, dtype=torch.float64, device=grad.device)
    extra[:-1] = scalars
    extra[-1] = 1.0 if active else 0.0
    if world > 1:
        dist.all_reduce(grad, op=dist.ReduceOp.SUM, group=group)
        dist.all_reduce(extra, op=dist.ReduceOp.SUM, group=group)
    scalars.copy_(extra[:-1])
    grad.div_(extra[-1].clamp(min=1.0))
    return extra[-1]
