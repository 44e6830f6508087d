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

    p2p = None        # P2PExchange when the one-shot exchange is up (set by the learner)

    def all_reduce_mean_(self, flat):
        if is_distributed():
            if self.p2p is not None and flat.is_cuda and flat.numel() <= self.p2p.n_floats:
                return self.p2p.allreduce_mean_(flat)
            dist.all_reduce(flat, op=dist.ReduceOp.SUM)
            flat.div_(dist.get_world_size())
        return flat

    def broadcast_(self, flat, src=0):
        if is_distributed():
            dist.broadcast(flat, src=src)
        return flat


def warm_up_collective(device):
    """One throw-away all-reduce on a side stream before a collective is captured into a HIP graph (PyTorch's rule for
    graph capture: whatever a first call sets up -- communicator, streams, buffers -- must exist before capture begins)."""
    if not (dist.is_available() and dist.is_initialized()):
        return
    mode = os.environ.get('MGP_WARMUP', 'side')
    if mode == 'none':
        return
    if mode == 'side':
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            dist.all_reduce(torch.zeros((8,), device=device), op=dist.ReduceOp.SUM)
        torch.cuda.current_stream().wait_stream(side)
    else:
        dist.all_reduce(torch.zeros((8,), device=device), op=dist.ReduceOp.SUM)
    torch.cuda.synchronize()


def _comm_device():
    """Device collectives run on: the rank's GPU under RCCL ("nccl"), the host under gloo."""
    if dist.get_backend() == 'nccl':
        return torch.device('cuda', torch.cuda.current_device())
    return torch.device('cpu')


def any_rank(flag):
    """True on every rank if `flag` is true on any (one MAX all-reduce of a single int on the collective device): how the
    ranks of a data-parallel round agree on a failure only some of them observed."""
    if not is_distributed():
        return bool(flag)
    t = torch.tensor([1 if flag else 0], dtype=torch.int32, device=_comm_device())
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return bool(int(t.item()))


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


class P2PExchange(object):
    """One-shot exchange of the flat gradient between the ranks of one node (libmgp's mgp_p2p_*: IPC-mapped mailboxes,
    64-bit {sequence | fp32} packets pushed into every peer's memory, summed in rank order -- bit-identical on every rank;
    csrc/p2p_device.h).  Replaces the ring all-reduce of `dist.all_reduce` for the 6.9 KB gradient of a data-parallel
    DAGGER update (one hop instead of 2(W-1)) and lets the exchange run INSIDE the update's second kernel
    (mgp_train_step_p2p).  Built on top of an initialised process group, which only carries the 64-byte IPC handles and
    the bring-up check; ranks may share a device (IPC between processes), which is how the one-GPU test box runs it.

    `P2PExchange.create(n_floats)` returns None when the exchange cannot be brought up (MGP_P2P=0, world > 8, IPC refused,
    or the self-test failing on any rank): callers then keep the torch.distributed collective."""

    last_bringup = None   # why the last create() on this rank returned what it did: {'ok', 'stage', 'local_ok', 'world'}

    def __init__(self, handle, world, rank, n_floats, mem_kind):
        self.handle, self.world, self.rank, self.n_floats, self.mem_kind = handle, world, rank, n_floats, mem_kind

    @classmethod
    def create(cls, n_floats, device=None, self_test=True):
        import ctypes
        from . import _lib
        if not is_distributed():                      # a property of the process group: the same answer on every rank
            return None
        world, rk = dist.get_world_size(), dist.get_rank()
        if world < 2:
            return None
        L = _lib.lib()
        cdev = _comm_device()
        ok = torch.ones((1,), dtype=torch.int32, device=cdev)
        ptr, comm = ctypes.c_void_p(), None
        hb = L.mgp_p2p_handle_bytes()
        mine = torch.zeros((hb,), dtype=torch.uint8)
        # rank-LOCAL preconditions (environment, device, allocation, handle export) only clear this rank's `ok`; every rank
        # then still walks through every collective below and the MIN all-reduce turns one refusal into everyone's None
        local_ok = os.environ.get('MGP_P2P', '1') != '0' and torch.cuda.is_available() and world <= 8
        if local_ok:
            # the mailbox and the control words are allocated on the CURRENT HIP device: that must be the rank's own
            want = torch.device(device).index if device is not None and torch.device(device).index is not None \
                else torch.cuda.current_device()
            with torch.cuda.device(want):
                local_ok = L.mgp_p2p_create(world, rk, int(n_floats), ctypes.byref(ptr)) == 0
        if local_ok:
            raw = (ctypes.c_ubyte * hb)()
            if L.mgp_p2p_handle(ptr, raw) == 0:
                mine = torch.tensor(list(raw), dtype=torch.uint8)
            else:
                local_ok = False
        stage = 'local preconditions / mailbox allocation / handle export'
        cls.last_bringup = dict(ok=False, stage=stage, local_ok=bool(local_ok), world=world)
        if not local_ok:
            ok.zero_()
        # every rank takes part in every collective below, whatever happened locally (no rank may be left waiting)
        parts = [torch.zeros((hb,), dtype=torch.uint8, device=cdev) for _ in range(world)]
        dist.all_gather(parts, mine.to(cdev))
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if int(ok.item()) == 1:
            stage = 'hipIpcOpenMemHandle of the peers\' mailboxes (mgp_p2p_connect)'
            blob = bytes(torch.cat([p_.cpu() for p_ in parts]).tolist())
            local_ok = L.mgp_p2p_connect(ptr, blob) == 0
            if not local_ok:
                ok.zero_()
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if int(ok.item()) == 1:
            stage = 'self-test (two exchanges of known values)'
            kind = ctypes.c_int()
            L.mgp_p2p_info(ptr, None, None, None, ctypes.byref(kind))
            comm = cls(ptr, world, rk, int(n_floats), kind.value)
            local_ok = (not self_test) or comm._self_test(device)
            if not local_ok:
                ok.zero_()
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if int(ok.item()) != 1:
            # refused somewhere: say so ONCE per rank that saw it locally (callers keep the torch.distributed collective)
            cls.last_bringup = dict(ok=False, stage=stage, local_ok=bool(local_ok), world=world)
            if not local_ok and os.environ.get('MGP_P2P', '1') != '0':
                import warnings
                warnings.warn("one-shot gradient exchange not available on rank %d of %d: failed at %s; falling back to the "
                              "torch.distributed all-reduce" % (rk, world, stage), RuntimeWarning)
            if ptr.value:
                L.mgp_p2p_destroy(ptr)
            return None
        cls.last_bringup = dict(ok=True, stage='up', local_ok=True, world=world)
        return comm

    def _self_test(self, device=None):
        """Two exchanges of known values with a short timeout: both slots, every entry, every peer."""
        from . import _lib
        L = _lib.lib()
        dev = torch.device('cuda', torch.cuda.current_device()) if device is None else torch.device(device)
        # MGP_P2P_TIMEOUT_MS: how long an exchange waits for a peer before it reports an error (default 5 s; 2 s in this
        # self-test).  Ranks that SHARE a device (tests, dry runs) take turns on it and need more on a loaded box.
        import os
        t_ms = int(os.environ.get('MGP_P2P_TIMEOUT_MS', '0'))
        L.mgp_p2p_set_timeout_ms(self.handle, max(2000, t_ms))
        good = True
        dist.barrier()
        for rep in range(2):
            buf = (torch.arange(self.n_floats, device=dev, dtype=torch.float32) * 0.5 + (self.rank + 1) * (rep + 1))
            self.allreduce_mean_(buf)
            # the kernel adds in rank order; this reference adds the same values in the same order
            ref = torch.zeros_like(buf)
            for q in range(self.world):
                ref += torch.arange(self.n_floats, device=dev, dtype=torch.float32) * 0.5 + (q + 1) * (rep + 1)
            ref /= self.world
            good = good and bool(torch.equal(buf, ref))
        good = good and self.status()[0] == 0
        L.mgp_p2p_set_timeout_ms(self.handle, max(5000, t_ms))
        return good

    def allreduce_mean_(self, flat):
        """flat <- mean over ranks, in place, one launch on the current stream (graph capturable)."""
        from . import _lib, ops
        assert flat.is_cuda and flat.dtype == torch.float32 and flat.is_contiguous() and flat.numel() <= self.n_floats
        _lib.check(_lib.lib().mgp_p2p_allreduce_mean(self.handle, ops._ptr(flat), flat.numel(), ops._stream()),
                   'mgp_p2p_allreduce_mean')
        return flat

    def status(self):
        """(status, exchanges completed) after synchronising the current stream; status != 0: a peer was late (error)."""
        import ctypes
        from . import _lib, ops
        st, seq = ctypes.c_int(), ctypes.c_int()
        _lib.check(_lib.lib().mgp_p2p_status(self.handle, ctypes.byref(st), ctypes.byref(seq), ops._stream()), 'mgp_p2p_status')
        return st.value, seq.value

    def check(self):
        st, _ = self.status()
        if st != 0:
            from ._lib import MgpError
            raise MgpError("one-shot gradient exchange: a peer did not publish within the timeout (rank %d of %d)"
                           % (self.rank, self.world))

    def close(self):
        from . import _lib
        if self.handle is not None and self.handle.value:
            _lib.lib().mgp_p2p_destroy(self.handle)
        self.handle = None
