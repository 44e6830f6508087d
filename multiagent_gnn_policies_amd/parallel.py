"""Multi-GPU plumbing: one process per MI355X, torch.distributed (backend "nccl" == RCCL over xGMI).

Episodes are independent (reference train.py:18: one env per experiment), so rollouts shard with no
data-path collective.  The only exchange in DAGGER training is the gradient of the 1,730-parameter
Actor: ONE flat fp32 buffer (6,920 bytes) all-reduced per update -- latency-bound, so it is a single
in-place collective on a persistent buffer, never per-tensor.
"""
import os

import torch
import torch.distributed as dist


def is_distributed():
    return dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1


def rank():
    return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0


def world_size():
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def init_from_env(backend=None):
    """Initialise torch.distributed from RANK / WORLD_SIZE / MASTER_* (torchrun contract).
    Returns (rank, world, local_rank).  No-op for a single process."""
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rk = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if world > 1 and not (dist.is_available() and dist.is_initialized()):
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29500')
        if backend is None:
            # MGP_DIST_BACKEND=gloo lets several ranks share ONE GPU (tests on a 1-GPU box); production = RCCL
            backend = os.environ.get('MGP_DIST_BACKEND') or ('nccl' if torch.cuda.is_available() else 'gloo')
        if torch.cuda.is_available():
            torch.cuda.set_device(local_device_index(local))
        dist.init_process_group(backend=backend, rank=rk, world_size=world)
    return rk, world, local


def local_device_index(local_rank=None):
    """GPU index of this rank: LOCAL_RANK, folded onto the visible devices (several ranks may share a GPU in tests)."""
    if local_rank is None:
        local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    n = torch.cuda.device_count() if torch.cuda.is_available() else 1
    return local_rank % max(1, n)


def shard_range(n_items, rk=None, world=None):
    """Contiguous block partition of `n_items` independent episodes: rank r gets [lo, hi)."""
    rk = rank() if rk is None else rk
    world = world_size() if world is None else world
    base, rem = divmod(n_items, world)
    lo = rk * base + min(rk, rem)
    return lo, lo + base + (1 if rk < rem else 0)


class FlatGradSync(object):
    """All-reduce (mean) of one flat gradient buffer + one-off parameter broadcast."""

    def all_reduce_mean_(self, flat):
        if is_distributed():
            dist.all_reduce(flat, op=dist.ReduceOp.SUM)
            flat.div_(dist.get_world_size())
        return flat

    def broadcast_(self, flat, src=0):
        if is_distributed():
            dist.broadcast(flat, src=src)
        return flat


def all_gather_floats(values):
    """Gather a python list of floats from every rank (episode rewards for mean/std)."""
    if not is_distributed():
        return list(values)
    out = [None] * dist.get_world_size()
    dist.all_gather_object(out, list(values))
    return [v for part in out for v in part]
