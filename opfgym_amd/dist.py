"""Multi-GPU: one process per GPU, the batch sharded over ranks.

Instances are independent (the reference has no cross-environment state at
all: one OpfEnv per net, opf_env.py:58), so the data path needs NO collective:
every rank resets/steps its own contiguous shard.  The only exchange is the
optional re-assembly of per-instance outputs (reward, flags, observations) for
a learner that wants the full batch on every rank: one all-gather per tensor
over `torch.distributed` (backend "nccl" = RCCL over xGMI on the GPU box,
"gloo" in the CPU tests).
"""
from __future__ import annotations

import os


_FORCE_COLLECTIVE = False


def force_collective() -> bool:
    """True after `init_from_env(..., force_collective=True)`: a world of ONE rank has initialised its process group and
    runs every collective of this module for real instead of short-circuiting — the way to execute the RCCL code path
    (communicator set-up, `all_gather_into_tensor` on device tensors, the asynchronous gather, the barrier and the
    max-reduce of the bench) on a box with a single GPU.  An argument of the caller (bench.py maps its own
    OPFX_FORCE_COLLECTIVE switch onto it); this module reads the torchrun variables and nothing else."""
    return _FORCE_COLLECTIVE


def _local_only(world: int) -> bool:
    import torch.distributed as dist
    return not dist.is_initialized() or (world == 1 and not force_collective())


def init_from_env(backend=None, force_collective=False):
    """Initialise torch.distributed from RANK/WORLD_SIZE/MASTER_* (torchrun).  `backend`: None = 'nccl' (RCCL) with a GPU,
    else 'gloo'.  `force_collective`: initialise the group even for a world of one (see `force_collective()`)."""
    import torch
    import torch.distributed as dist
    global _FORCE_COLLECTIVE
    _FORCE_COLLECTIVE = bool(force_collective)
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if (world == 1 and not _FORCE_COLLECTIVE) or dist.is_initialized():
        return int(os.environ.get('RANK', '0')), world, int(os.environ.get('LOCAL_RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if backend is None:
        backend = 'nccl' if torch.cuda.is_available() else 'gloo'
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('MASTER_PORT', '29511')
    if backend == 'nccl':
        # one process per GPU: bind the device before the communicator is created
        torch.cuda.set_device(local_rank)
        dist.init_process_group(backend=backend, device_id=torch.device('cuda', local_rank))
    else:
        dist.init_process_group(backend=backend)
    return dist.get_rank(), dist.get_world_size(), local_rank


def shard_bounds(total: int, rank: int, world: int):
    """Contiguous shard [lo, hi) of `total` instances for `rank` (sizes differ by
    at most one; whole instances only, so the N-1 reduction over the
    contingencies of an instance never crosses ranks)."""
    base, rem = divmod(int(total), int(world))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def all_gather_rows(local, world: int, sizes=None):
    """Concatenate per-rank [b_r, ...] tensors along dim 0 on every rank.
    Equal shard sizes use one all_gather_into_tensor; ragged shards pad to the
    largest shard and trim."""
    import torch
    import torch.distributed as dist
    if _local_only(world):
        return local
    if local.is_cuda and dist.get_backend() == 'gloo':
        # debugging aid (several ranks sharing one GPU): stage through the host
        return all_gather_rows(local.cpu(), world, sizes).to(local.device)
    if sizes is None or len(set(sizes)) == 1:
        out = torch.empty((local.shape[0] * world,) + tuple(local.shape[1:]), dtype=local.dtype,
                          device=local.device)
        dist.all_gather_into_tensor(out, local.contiguous())
        return out
    m = max(sizes)
    pad = torch.zeros((m,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[:local.shape[0]] = local
    out = torch.empty((m * world,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(out, pad)
    return torch.cat([out[r * m:r * m + sizes[r]] for r in range(world)], dim=0)


class OverlappedGather:
    """The per-step all-gather of one output WITHOUT stalling the producer: the local rows are copied into one of
    two staging buffers and gathered asynchronously — on RCCL's own stream, behind the copy — while the next step's
    kernel already runs; `submit` hands back the full-batch tensor of the PREVIOUS step (complete by then), `flush`
    the last one.  A learner consumes step k's batch while step k+1 is simulated, which is how the gather comes off
    the critical path (a blocking gather would add its latency to every step: ~10 % at 0.27 ms per step).

    `sizes`: rows per rank when the shards are ragged (`shard_bounds` with a batch the world size does not divide,
    e.g. 65 536 instances on 3, 6 or 7 GPUs): every rank then stages a shard padded to the largest one — the
    collective needs equal counts — and the padding rows are cut out on hand-over.

    LIFETIME of what `submit` / `flush` return: with equal shards it is one of the two persistent gather buffers and
    is overwritten by the submit after next (use it before, or pass `clone=True`); with ragged shards it is a fresh
    tensor (the trim copies)."""

    def __init__(self, world: int, sizes=None, clone: bool = False):
        self.world, self.clone = world, clone
        self.sizes = None if sizes is None or len(set(sizes)) == 1 else [int(v) for v in sizes]
        if self.sizes is not None and len(self.sizes) != world:
            raise ValueError(f'OverlappedGather: {len(self.sizes)} shard sizes for a world of {world}')
        self._stage, self._out, self._work, self._k = [None, None], [None, None], [None, None], 0

    def _rows(self, local):
        import torch.distributed as dist
        if self.sizes is None:
            return local.shape[0]
        mine = self.sizes[dist.get_rank()]
        if local.shape[0] != mine:
            raise ValueError(f'OverlappedGather: rank {dist.get_rank()} submitted {local.shape[0]} rows, its shard has {mine}')
        return max(self.sizes)

    def _hand_over(self, out):
        import torch
        if out is None:
            return None
        if self.sizes is not None:
            m = max(self.sizes)
            return torch.cat([out[r * m:r * m + self.sizes[r]] for r in range(self.world)], dim=0)
        return out.clone() if self.clone else out

    def submit(self, local):
        import torch
        import torch.distributed as dist
        if _local_only(self.world):
            return local
        if local.is_cuda and dist.get_backend() == 'gloo':        # (debugging aid: ranks sharing one GPU)
            prev, self._last = getattr(self, '_last', None), all_gather_rows(local, self.world, self.sizes)
            return prev
        slot, other = self._k & 1, (self._k & 1) ^ 1
        self._k += 1
        m = self._rows(local)
        if self._stage[slot] is None:
            self._stage[slot] = torch.zeros((m,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
            self._out[slot] = torch.empty((m * self.world,) + tuple(local.shape[1:]), dtype=local.dtype,
                                          device=local.device)
        if self._work[slot] is not None:
            self._work[slot].wait()                            # (two steps old: long finished)
        self._stage[slot][:local.shape[0]].copy_(local)
        self._work[slot] = dist.all_gather_into_tensor(self._out[slot], self._stage[slot], async_op=True)
        if self._work[other] is None:
            return None
        self._work[other].wait()
        return self._hand_over(self._out[other])

    def flush(self):
        """The full-batch tensor of the last submitted step."""
        if _local_only(self.world):
            return None
        if getattr(self, '_last', None) is not None:
            return self._last
        slot = (self._k - 1) & 1
        if self._k == 0 or self._work[slot] is None:
            return None
        self._work[slot].wait()
        return self._hand_over(self._out[slot])


class ShardedBatch:
    """Wraps a per-rank environment factory: `total_batch` instances split over
    the ranks; `step()` returns the local shard's outputs plus, for the names in
    `gather`, full-batch tensors assembled with one all-gather each."""

    def __init__(self, make_env, total_batch, rank, world, gather=('reward',)):
        self.rank, self.world = rank, world
        self.lo, self.hi = shard_bounds(total_batch, rank, world)
        self.sizes = [shard_bounds(total_batch, r, world)[1] - shard_bounds(total_batch, r, world)[0]
                      for r in range(world)]
        self.env = make_env(self.hi - self.lo)
        self.gather = tuple(gather)

    def reset(self, **kw):
        return self.env.reset(**kw)

    def step(self, local_actions):
        obs, reward, term, trunc, info = self.env.step(local_actions)
        full = {}
        named = {'reward': reward, 'obs': obs, 'terminated': term.to(reward.dtype),
                 'cost': info['cost']}
        for name in self.gather:
            full[name] = all_gather_rows(named[name], self.world, self.sizes)
        return (obs, reward, term, trunc, info), full
