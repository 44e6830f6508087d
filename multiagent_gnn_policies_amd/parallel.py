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
    force = os.environ.get('MGP_FORCE_DIST') == '1'          # world-size-1 process group (RCCL bring-up on one GPU)
    if (world > 1 or force) and not (dist.is_available() and dist.is_initialized()):
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


def _comm_device():
    """Device collectives run on: the rank's GPU under RCCL ("nccl"), the host under gloo."""
    if dist.get_backend() == 'nccl':
        return torch.device('cuda', torch.cuda.current_device())
    return torch.device('cpu')


def all_gather_floats(values):
    """Gather a list of floats (episode rewards for mean / std, reference gnn_dagger.py:235-237) from every rank, rank
    order preserved.  Two fixed-shape tensor collectives -- the counts, then the values padded to the longest list -- as
    fp64, so the statistics equal a single-process run's bit for bit; no pickling (all_gather_object would serialise
    through a byte tensor and, under RCCL, bounce it through the GPU twice)."""
    if not is_distributed():
        return list(values)
    dev, world = _comm_device(), dist.get_world_size()
    n = torch.tensor([len(values)], dtype=torch.int64, device=dev)
    counts = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(counts, n)
    counts = [int(c.item()) for c in counts]
    width = max(max(counts), 1)
    mine = torch.zeros((width,), dtype=torch.float64, device=dev)
    if len(values):
        mine[:len(values)] = torch.tensor([float(v) for v in values], dtype=torch.float64)
    parts = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(parts, mine)
    return [float(v) for part, c in zip(parts, counts) for v in part[:c].cpu().tolist()]
