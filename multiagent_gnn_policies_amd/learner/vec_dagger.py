"""Vectorised, device-resident DAGGER: many episodes per GPU, nothing leaves HBM between the simulator, the state
update, the policy and the replay memory.  An extension of reference learner/gnn_dagger.py:126-243 (which steps ONE
environment through Python objects); the schedule is the reference's, applied per episode:

  * episode e (global index over rounds, ranks and lanes) uses the reference's running-product beta_e (:148, BetaSchedule)
  * every step stores (state, expert label) and steps the env with the expert w.p. beta_e else the policy  (:156-178)
  * after a round of `n_envs` episodes: `updates_per_step` updates per episode of the round            (:182-188)
  * final statistics = mean / std of `n_test_episodes` policy-only episode rewards                     (:221-237)

`DeviceReplay` keeps the reference's ring + sample-without-replacement semantics (replay_buffer.py:6-49, Python's
`random` RNG) but stores dense states in HBM: 10,000 transitions of N=100,K=3 are 1.3 GB of 288 GB.
Ranks shard episodes (parallel.py); the only collective is the flat-gradient all-reduce inside gradient_step.
"""
import random

import numpy as np
import torch

from .. import ops, parallel
from ..envs import FlockParams, VecFlock
from ..envs.flocking import _REGISTRY, sample_initial_states, use_grid
from .rollouts import policy_rollout
from .gnn_dagger import DAGGER, BetaSchedule
from .state_with_delay import BatchedDelayState


class DeviceReplay(object):
    """Ring buffer of transitions resident on the device (state = delay_state + delay_gso, label = expert action)."""

    def __init__(self, max_size, K, F, N, n_a, device):
        self.max_size = max_size
        kw = dict(device=device, dtype=torch.float32)
        self.delay_state = torch.empty((max_size, K, F, N), **kw)
        self.delay_gso = torch.empty((max_size, K, N, N), **kw)
        self.action = torch.empty((max_size, 1, n_a, N), **kw)
        self.curr_size = 0
        self.position = 0
        self.device = device

    def insert_batch(self, delay_state, delay_gso, action):
        """Append B transitions (oldest overwritten), same order as B consecutive `insert` calls."""
        B = delay_state.shape[0]
        idx = (torch.arange(B, device=self.device) + self.position) % self.max_size
        self.delay_state.index_copy_(0, idx, delay_state)
        self.delay_gso.index_copy_(0, idx, delay_gso)
        self.action.index_copy_(0, idx, action)
        self.position = (self.position + B) % self.max_size
        self.curr_size = min(self.max_size, self.curr_size + B)

    def sample(self, num_samples, out=None):
        """Without replacement, Python `random` RNG (reference replay_buffer.py:40).  `out` = (X, G, Y) destination
        tensors (e.g. the static buffers of the HIP-graph update): the gather is then the only copy of the batch."""
        ids = random.sample(range(self.curr_size), num_samples)
        idx = torch.tensor(ids, device=self.device, dtype=torch.long)
        if out is not None:
            torch.index_select(self.delay_state, 0, idx, out=out[0])
            torch.index_select(self.delay_gso, 0, idx, out=out[1])
            torch.index_select(self.action, 0, idx, out=out[2])
            return out
        return (self.delay_state.index_select(0, idx), self.delay_gso.index_select(0, idx),
                self.action.index_select(0, idx))

    def clear(self):
        self.curr_size = 0
        self.position = 0


UPDATES_PER_GRAPH = 32


def _dp_mode(learner, frames):
    """How a data-parallel round of graph-captured updates exchanges its gradients:
      'p2p'   the one-shot exchange inside the update's second launch (mgp_train_step_p2p); any process-group backend
      'rccl'  mgp_train_grads -> dist.all_reduce captured in the graph -> mgp_adam_step_filed (frame replay only: the
              gradient kernel takes its minibatch from the gathered slots)
      None    neither (gloo without the exchange): updates are enqueued one by one by the caller"""
    import os
    import torch.distributed as dist
    if getattr(learner, 'p2p', None) is not None and learner.p2p.n_floats > learner.actor_optim.flat.numel():
        return 'p2p'
    if frames and dist.get_backend() == 'nccl' and os.environ.get('MGP_DIST_GRAPH', '1') != '0':
        return 'rccl'
    return None


def _replay_updates(obj, U):
    """Run U consecutive updates of `obj` (IndexedUpdates / FrameUpdates: everything an update needs -- minibatch indices,
    cursor, step counter, losses -- lives on the device).  One update is 2-3 short launches (~30 us of GPU time), less than
    the host spends on one graph replay; so UPDATES_PER_GRAPH updates are captured back to back in ONE graph, and a round of
    thousands of updates is GPU-bound instead of host-bound (60 -> ~32 us per update).  The remainder runs on a one-update graph."""
    if obj.graph is None:
        torch.cuda.synchronize()
        obj.graph = torch.cuda.CUDAGraph()
        with ops.graph_capture(obj.graph):
            obj._enqueue()
        obj.graph_many = torch.cuda.CUDAGraph()
        with ops.graph_capture(obj.graph_many):
            if hasattr(obj, '_enqueue_many'):
                obj._enqueue_many(UPDATES_PER_GRAPH)
            else:
                for _ in range(UPDATES_PER_GRAPH):
                    obj._enqueue()
        # capture does not execute: the cursor / step counter / weights are untouched
    if U <= 0:
        return
    full, rest = divmod(U, UPDATES_PER_GRAPH)
    for _ in range(full):
        obj.graph_many.replay()
    for _ in range(rest):
        obj.graph.replay()


def _run_sampled(obj, U, sampler, batch_sampler=None):
    """U updates whose minibatch indices come from `sampler()` (one list of batch_size indices per call: the reference's
    `random.sample` per update) or `batch_sampler(n)` ((n, batch_size) int64 array: the same draws, sample_batch), pipelined:
    the host draws the indices of a graph of UPDATES_PER_GRAPH updates into its own rows of a pinned staging table, enqueues
    their upload and the graph's replay, and goes on to the next graph without waiting for anything -- every graph has its own
    staging rows, so the host is through the whole round (about 1.5 us per update) long before the GPU is, and whatever it
    does next (train_dagger_vec: drawing the next round's reset states) runs under the GPU's updates.  Returns the sum of
    the losses (device tensor; reading it is the caller's synchronisation point)."""
    assert 0 < U <= obj.cap
    assert obj.idx.shape[0] >= min(obj.cap, U) and obj.idx.shape[1] == obj.B
    rows = ((obj.cap + UPDATES_PER_GRAPH - 1) // UPDATES_PER_GRAPH) * UPDATES_PER_GRAPH
    if getattr(obj, '_stage', None) is None or obj._stage.shape[0] < rows:
        obj._stage = torch.empty((rows, obj.B), dtype=torch.long).pin_memory()
        obj._stage_done = None
    if obj._stage_done is not None:
        obj._stage_done.synchronize()                             # the previous round's uploads have left the staging table
    obj.cursor.zero_()
    _replay_updates(obj, 0)                                       # make sure both graphs exist
    done = 0
    while done < U:
        n = min(UPDATES_PER_GRAPH, U - done)
        buf = obj._stage[done:done + n]
        if batch_sampler is not None:                             # (n, B) int64 array in one call (sample_batch: same stream, same values)
            buf.copy_(torch.from_numpy(batch_sampler(n)))
        else:
            buf.copy_(torch.tensor([sampler() for _ in range(n)], dtype=torch.long))
        obj.idx[done:done + n].copy_(buf, non_blocking=True)
        # a graph of UPDATES_PER_GRAPH updates maps update c to gather slot c % UPDATES_PER_GRAPH: it must start on a multiple
        assert done % UPDATES_PER_GRAPH == 0
        _replay_updates(obj, n)
        done += n
    obj._stage_done = torch.cuda.Event()
    obj._stage_done.record()
    obj.learner.actor_optim.step_count += U
    return obj.loss_hist[:U].sum()


_SAMPLE_BATCH_OK = {}          # k -> the block form reproduces random.sample on this interpreter (checked on its first real use)


def _sample_setsize(k):
    """random.sample's set-size threshold (CPython: 21, plus 4 ** ceil(log4(3 k)) for k > 5): populations up to it take the
    pool branch of random.sample, which the block form does not restate."""
    import math
    return 21 + (4 ** math.ceil(math.log(k * 3, 4)) if k > 5 else 0)


def _sample_batch_sequential(n, k, count):
    return np.array([random.sample(range(n), k) for _ in range(count)], dtype=np.int64).reshape(max(count, 0), k)


def sample_batch(n, k, count):
    """The reference's minibatch draws (replay_buffer.py:40) for `count` updates at once.  The block form below re-implements
    CPython internals (random.sample's set-size threshold, _randbelow's rejection loop, getrandbits' word order): on first use
    for a shape it is compared with random.sample itself -- values AND generator state, on a saved and restored state -- and
    an interpreter on which they differ keeps the sequential sampler."""
    if count <= 0 or k <= 0 or n <= _sample_setsize(k) or n >= (1 << 32):
        return _sample_batch_sequential(n, k, count)              # the block form does not apply: nothing to verify, nothing cached
    ok = _SAMPLE_BATCH_OK.get(k)                                  # (the block path is taken for every n beyond the threshold)
    if ok is None:
        state = random.getstate()
        try:
            a = _sample_batch_block(n, k, 8)
            sa = random.getstate()
            random.setstate(state)
            b = _sample_batch_sequential(n, k, 8)
            ok = bool(np.array_equal(a, b)) and sa == random.getstate()
        except Exception:
            ok = False
        finally:
            random.setstate(state)
        _SAMPLE_BATCH_OK[k] = ok
    return _sample_batch_block(n, k, count) if ok else _sample_batch_sequential(n, k, count)


def _sample_batch_block(n, k, count):
    """`count` consecutive draws of random.sample(range(n), k) -- the reference's minibatch sampler (replay_buffer.py:40) -- as one
    (count, k) int64 array: the SAME values from the SAME stream of Python's global generator, which is left in the state
    `count` calls of random.sample would leave it in.  A round of updates draws thousands of minibatches; at ~10 us per
    random.sample call the host was as slow as the GPU's 11.7 us per update.  How: random.sample (n larger than its set-size
    threshold) is k draws of _randbelow(n) with repeats inside a minibatch rejected, _randbelow is getrandbits(bits(n)) until
    the value is below n, and getrandbits(b <= 32) is the top b bits of one MT19937 word -- so a block of words fetched at
    once (getrandbits(32 M): M successive words, little end first) holds every value the calls would see, in order; the block
    is filtered with numpy, and the generator is rewound and advanced by exactly the number of words the calls would have
    consumed.  Minibatches with an internal repeat (rare: k^2 / 2n) are finished by the sequential rule on the same stream."""
    if count <= 0 or k <= 0 or n <= _sample_setsize(k) or n >= (1 << 32):
        return _sample_batch_sequential(n, k, count)
    bits = n.bit_length()
    need = count * k
    state = random.getstate()
    M = int(need * ((1 << bits) / n) * 1.05) + 64
    while True:
        words = np.frombuffer(random.getrandbits(32 * M).to_bytes(4 * M, 'little'), dtype='<u4')
        r = words >> np.uint32(32 - bits)
        ok = np.flatnonzero(r < n)
        out = None
        if ok.size >= need:
            acc = r[ok[:need]].astype(np.int64).reshape(count, k)
            srt = np.sort(acc, axis=1)
            dup = np.flatnonzero((srt[:, 1:] == srt[:, :-1]).any(axis=1))
            if dup.size == 0:
                out, consumed = acc, int(ok[need - 1]) + 1
            else:
                # sequential rule from the first minibatch with a repeat on: accepted values in stream order, repeats skipped
                g0 = int(dup[0])
                vals = r[ok].tolist()
                ptr, rows, short = g0 * k, [], False
                for _ in range(g0, count):
                    seen, row = set(), []
                    while len(row) < k:
                        if ptr >= len(vals):
                            short = True
                            break
                        v = vals[ptr]; ptr += 1
                        if v not in seen:
                            seen.add(v); row.append(v)
                    if short:
                        break
                    rows.append(row)
                if not short:
                    out = np.concatenate([acc[:g0], np.array(rows, dtype=np.int64).reshape(-1, k)], axis=0)
                    consumed = int(ok[ptr - 1]) + 1
        random.setstate(state)
        if out is not None:
            random.getrandbits(32 * consumed)                    # advance by exactly the words the calls would have drawn
            return out
        M *= 2


class IndexedUpdates(object):
    """A round of DAGGER updates with no per-update host work: the minibatch indices of the whole round are uploaded once,
    and every update is one replay of a two-launch HIP graph (mgp_train_step_indexed) that gathers its batch from the
    replay arrays inside the kernel, applies Adam, advances the device-side update cursor and files its loss.
    Sampling is the reference's (`random.sample` without replacement per update, replay_buffer.py:40)."""

    def __init__(self, learner, memory, batch_size, max_updates):
        import ctypes
        from .. import _lib
        actor, opt = learner.actor, learner.actor_optim
        dev = opt.flat.device
        dims = tuple(actor.layers)
        self.cdims = (ctypes.c_int * len(dims))(*dims)
        self.nl, self.K, self.N, self.B = actor.n_layers, actor.k, learner.n_agents, batch_size
        self.learner, self.memory, self.cap = learner, memory, max_updates
        L = _lib.lib()
        self.idx = torch.zeros((max_updates, batch_size), device=dev, dtype=torch.long)
        self.cursor = torch.zeros((1,), device=dev, dtype=torch.int32)
        self.loss_hist = torch.zeros((max_updates,), device=dev, dtype=torch.float32)
        self.step_dev = opt.step_dev                # FlatAdam's shared device step counter
        self.ws = torch.zeros((L.mgp_train_workspace(self.cdims, self.nl, batch_size, self.K, self.N),), device=dev)
        self.graph = None

    @staticmethod
    def supported(learner, batch_size, N, frames=False):
        import ctypes
        from .. import _lib
        if not (learner.use_graphed_update and learner.use_train_step):
            return False
        if parallel.is_distributed() and _dp_mode(learner, frames) is None:
            return False
        dims = tuple(learner.actor.layers)
        cd = (ctypes.c_int * len(dims))(*dims)
        return learner.actor.ind_agg == 0 and bool(_lib.lib().mgp_train_supported(cd, learner.actor.n_layers, batch_size,
                                                                                 learner.actor.k, N))

    def _enqueue(self):
        from .. import _lib, ops
        L, o, m = _lib.lib(), self.learner.actor_optim, self.memory
        if parallel.is_distributed():                # data parallel: the one-shot exchange inside the second launch
            _lib.check(L.mgp_train_step_p2p(
                ops._ptr(m.delay_state), ops._ptr(m.delay_gso), ops._ptr(m.action), self.idx.data_ptr(), self.cursor.data_ptr(),
                ops._ptr(self.loss_hist), self.cap, ops._ptr(o.flat), ops._ptr(o.flat_grad), ops._ptr(o.m), ops._ptr(o.v),
                self.cdims, self.nl, o.lr, o.betas[0], o.betas[1], o.eps, self.step_dev.data_ptr(), None, ops._ptr(self.ws),
                self.B, self.K, self.N, self.learner.p2p.handle, ops._stream()), 'mgp_train_step_p2p')
            return
        _lib.check(L.mgp_train_step_indexed(
            ops._ptr(m.delay_state), ops._ptr(m.delay_gso), ops._ptr(m.action), self.idx.data_ptr(), self.cursor.data_ptr(),
            ops._ptr(self.loss_hist), self.cap, ops._ptr(o.flat), ops._ptr(o.flat_grad), ops._ptr(o.m), ops._ptr(o.v),
            self.cdims, self.nl, o.lr, o.betas[0], o.betas[1], o.eps, self.step_dev.data_ptr(), ops._ptr(self.ws),
            self.B, self.K, self.N, ops._stream()), 'mgp_train_step_indexed')

    def run_sampled(self, U, sampler=None):
        """U updates, indices drawn per update by `sampler` (default: the reference's random.sample over the replay rows),
        host sampling overlapped with the GPU (see _run_sampled)."""
        m = self.memory
        if sampler is None:
            return _run_sampled(self, U, None, batch_sampler=lambda n_: sample_batch(m.curr_size, self.B, n_))
        return _run_sampled(self, U, sampler)

    def run(self, ids):
        """ids: one list of `batch_size` replay rows per update.  Returns the sum of the updates' losses (device tensor)."""
        U = len(ids)
        assert 0 < U <= self.cap
        opt = self.learner.actor_optim
        self.idx[:U].copy_(torch.tensor(ids, dtype=torch.long), non_blocking=False)
        self.cursor.zero_()
        _replay_updates(self, U)
        opt.step_count += U
        return self.loss_hist[:U].sum()


class FrameReplay(object):
    """Compact device replay filled by the collecting rollout kernel (mgp_rollout_collect): a ring of FRAMES laid out
    [ring_steps][lanes] -- features x_t (6,N), membership bits of A_t (N x 2 u64), expert label (2,N), age -- 4.8 KB per
    transition at N = 100 where a dense (delay_state, delay_gso) pair takes 128 KB.  The K-tap state of a transition is
    rebuilt from its frame and its K - 1 predecessors in the ring (same lane) by mgp_replay_gather, so the ring keeps K - 1
    guard steps behind the sampled window.  Requires networks with SYMMETRIC membership (row n of the bits is used as column
    n of A_t; FlockParams.symmetric_network, checked by collect_supported).  Ring + sampling semantics are the reference's (replay_buffer.py:21-41): once
    full, the oldest transitions are overwritten; `sample_ids` draws without replacement from Python's `random` stream."""

    def __init__(self, lanes, capacity, K, N, device):
        self.lanes, self.K, self.N = lanes, K, N
        self.window_steps = max(1, (capacity + lanes - 1) // lanes)          # sampled window, in lock-step env steps
        self.ring_steps = self.window_steps + max(K - 1, 0)                  # + history guard
        S = self.ring_steps
        # 64-bit words per membership row: the collecting kernels' layouts (resident: 2 / 4; factored, N > 256: mgp_sparse_words)
        from .. import _lib
        self.nw = 2 if N <= 128 else (4 if N <= 256 else _lib.lib().mgp_sparse_words(N))
        self.feat = torch.zeros((S, lanes, 6, N), device=device, dtype=torch.float32)
        self.bits = torch.zeros((S, lanes, N, self.nw), device=device, dtype=torch.int64)
        # N > 256 (mgp_sparse_policy_collect): the row weights travel with the frame (the gather works from HBM, row by row)
        self.wrow = torch.zeros((S, lanes, N), device=device, dtype=torch.float32) if N > 256 else None
        self.label = torch.zeros((S, lanes, 2, N), device=device, dtype=torch.float32)
        self.age = torch.zeros((S, lanes), device=device, dtype=torch.int32)
        self.head = 0                 # ring step the next collected env step is filed at
        self._table_key, self._table = None, None
        self.steps_written = 0
        self.device = device

    @property
    def max_size(self):
        return self.window_steps * self.lanes

    @property
    def curr_size(self):
        return min(self.steps_written, self.window_steps) * self.lanes

    def bytes_per_transition(self):
        return (6 * self.N + 2 * self.N) * 4 + self.N * 8 * self.nw + 4 + (4 * self.N if self.wrow is not None else 0)

    def advance(self, T):
        """The collecting launch filed T env steps starting at ring step `head`."""
        self.head = (self.head + T) % self.ring_steps
        self.steps_written += T

    def frame_of(self, position):
        """Frame index (ring_step * lanes + lane) of the transition at buffer position `position` in [0, curr_size):
        positions run oldest to newest, lane-minor -- the order B consecutive `insert` calls per env step would give."""
        n_valid = min(self.steps_written, self.window_steps)
        step = (self.head - n_valid + position // self.lanes) % self.ring_steps
        return step * self.lanes + position % self.lanes

    def sample_ids(self, num_samples):
        """Frame indices of a minibatch: without replacement, Python `random` RNG (reference replay_buffer.py:40).  The
        position -> frame table is rebuilt only when the ring moved (once per collection round): this runs once per update
        on the host while the GPU replays the previous updates."""
        key = (self.head, self.steps_written)
        if self._table_key != key:
            n_valid = min(self.steps_written, self.window_steps)
            steps = (self.head - n_valid + np.arange(n_valid)) % self.ring_steps
            self._table_np = (steps[:, None] * self.lanes + np.arange(self.lanes)[None, :]).reshape(-1).astype(np.int64)
            self._table = self._table_np.tolist()
            self._table_key = key
        tbl = self._table
        return [tbl[i] for i in random.sample(range(len(tbl)), num_samples)]

    def sample_ids_many(self, num_samples, count):
        """`count` consecutive sample_ids(num_samples) as one (count, num_samples) int64 array: same values, same generator
        state afterwards (sample_batch)."""
        if count > 0:
            self.sample_ids(0)                                   # (re)builds the position -> frame table; draws nothing
        pos = sample_batch(len(self._table), num_samples, count)
        return self._table_np[pos]

    def sample(self, num_samples, out, mean_pooling=True):
        """Gather one minibatch into `out` = (X, G, Y) (the eager / data-parallel update path: one small H2D per update)."""
        idx = torch.tensor(self.sample_ids(num_samples), device=self.device, dtype=torch.long)
        from .. import ops
        ops.replay_gather(self, idx, out[0], out[1], out[2], mean_pooling)
        return out

    def clear(self):
        self.head, self.steps_written = 0, 0


class FrameUpdates(object):
    """A round of DAGGER updates on a FrameReplay with no per-update host work: the frame indices of the whole round are
    uploaded once; an update is mgp_replay_gather (rebuilds the minibatch's K-tap states from the frame ring at the
    device-side cursor) and the two launches of mgp_train_step_indexed on the gathered buffers (forward + MSE + backward per
    tile; reduction + Adam, which advances the cursor and files the loss).  The gathers do not depend on the weights: the
    graph of UPDATES_PER_GRAPH updates starts with ONE launch that gathers all of its minibatches (mgp_replay_gather_many,
    ~1 us per update instead of 9) into UPDATES_PER_GRAPH slots and each update reads its slot.

    `aggregated` (default wherever mgp_train_agg_supported says so; MGP_FRAME_AGG=0 or aggregated=False selects the dense
    form): the slots hold the AGGREGATED first-layer input Z (6 K, N) per sample instead of (X, G) -- mgp_replay_aggregate
    runs the K-hop products along the frames' bit rows, mgp_train_step_agg trains on the result.  Parameters are the only
    leaves of the reference's update (gnn_dagger.py:85-93), so the product's result is all it needs; the dense slices cost
    K N^2 floats per sample to write and read back (240 MB per minibatch at N = 1000)."""

    def __init__(self, learner, memory, batch_size, max_updates, mean_pooling, aggregated=None):
        import ctypes
        import os
        from .. import _lib
        actor, opt = learner.actor, learner.actor_optim
        dev = opt.flat.device
        dims = tuple(actor.layers)
        self.cdims = (ctypes.c_int * len(dims))(*dims)
        self.nl, self.K, self.N, self.B = actor.n_layers, actor.k, learner.n_agents, batch_size
        self.learner, self.memory, self.cap, self.mean_pooling = learner, memory, max_updates, mean_pooling
        L = _lib.lib()
        can_agg = bool(L.mgp_train_agg_supported(self.cdims, self.nl, batch_size, self.K, self.N)) and self.N <= 2048 and dims[0] == 6
        if aggregated is None:
            aggregated = can_agg and os.environ.get('MGP_FRAME_AGG', '1') != '0'
        assert can_agg or not aggregated
        self.aggregated = bool(aggregated)
        slots = UPDATES_PER_GRAPH * batch_size                      # one slot of batch_size samples per update of a graph
        if self.aggregated:
            self.Z = torch.zeros((slots, 6 * self.K, self.N), device=dev)
            self.X = self.G = None
        else:
            self.X = torch.zeros((slots, self.K, 6, self.N), device=dev)
            self.G = torch.zeros((slots, self.K, self.N, self.N), device=dev)
        self.Y = torch.zeros((slots, 1, actor.n_a, self.N), device=dev)
        self.idx = torch.zeros((max_updates + UPDATES_PER_GRAPH, batch_size), device=dev, dtype=torch.long)   # frame indices
        self.ident = torch.arange(batch_size, device=dev, dtype=torch.long).repeat(max_updates, 1).contiguous()
        # update number c of a round reads slot c % UPDATES_PER_GRAPH (full graphs start at multiples of it)
        self.ident_many = (self.ident + (torch.arange(max_updates, device=dev, dtype=torch.long) % UPDATES_PER_GRAPH)[:, None]
                           * batch_size).contiguous()
        self.cursor = torch.zeros((1,), device=dev, dtype=torch.int32)
        self.loss_hist = torch.zeros((max_updates,), device=dev, dtype=torch.float32)
        self.step_dev = opt.step_dev
        self.ws = torch.zeros((L.mgp_train_workspace(self.cdims, self.nl, batch_size, self.K, self.N),), device=dev)
        self.graph = None
        self.dp = _dp_mode(learner, True) if parallel.is_distributed() else None
        if self.dp == 'rccl':
            from .actor_fused import _ptr_array
            pv = opt.views(opt.flat)
            self.Wp, self.bp, self._keep = _ptr_array(pv[0::2]), _ptr_array(pv[1::2]), pv
            self.xbuf = torch.zeros((opt.flat.numel() + 1,), device=dev)       # gradient | loss: one all-reduce
            parallel.warm_up_collective(dev)

    @staticmethod
    def supported(learner, batch_size, N):
        import ctypes
        import os
        from .. import _lib
        if IndexedUpdates.supported(learner, batch_size, N, frames=True):
            return True
        # shapes only the aggregated form covers (the dense tile kernel also keeps K F N floats of X in LDS)
        if not (learner.use_graphed_update and learner.use_train_step) or os.environ.get('MGP_FRAME_AGG', '1') == '0':
            return False
        if parallel.is_distributed() and _dp_mode(learner, True) is None:
            return False
        dims = tuple(learner.actor.layers)
        cd = (ctypes.c_int * len(dims))(*dims)
        return (learner.actor.ind_agg == 0 and dims[0] == 6 and N <= 2048
                and bool(_lib.lib().mgp_train_agg_supported(cd, learner.actor.n_layers, batch_size, learner.actor.k, N)))

    def _train_agg(self, ident, slot):
        from .. import _lib, ops
        L, o = _lib.lib(), self.learner.actor_optim
        if self.dp == 'rccl':
            import torch.distributed as dist
            B, P = self.B, o.flat.numel()
            lo = slot * B
            _lib.check(L.mgp_train_grads_agg(ops._ptr(self.Z[lo:lo + B]), ops._ptr(self.Y[lo:lo + B]), self.Wp, self.bp,
                                             self.cdims, self.nl, ops._ptr(self.xbuf), ops._ptr(self.xbuf[P:]),
                                             ops._ptr(self.ws), B, self.K, self.N, ops._stream()), 'mgp_train_grads_agg')
            dist.all_reduce(self.xbuf, op=dist.ReduceOp.SUM)
            self.xbuf.div_(parallel.world_size())
            _lib.check(L.mgp_adam_step_filed(ops._ptr(o.flat), ops._ptr(self.xbuf), ops._ptr(o.m), ops._ptr(o.v), P, o.lr,
                                             o.betas[0], o.betas[1], o.eps, self.step_dev.data_ptr(), ops._ptr(self.xbuf[P:]),
                                             ops._ptr(self.loss_hist), self.cap, self.cursor.data_ptr(), ops._stream()),
                       'mgp_adam_step_filed')
            return
        comm = self.learner.p2p.handle if self.dp == 'p2p' else None
        _lib.check(L.mgp_train_step_agg(
            ops._ptr(self.Z), ops._ptr(self.Y), ident.data_ptr(), self.cursor.data_ptr(), ops._ptr(self.loss_hist), self.cap,
            ops._ptr(o.flat), ops._ptr(o.flat_grad), ops._ptr(o.m), ops._ptr(o.v), self.cdims, self.nl, o.lr, o.betas[0],
            o.betas[1], o.eps, self.step_dev.data_ptr(), None, ops._ptr(self.ws), self.B, self.K, self.N, comm,
            ops._stream()), 'mgp_train_step_agg')

    def _train(self, ident, slot=0):
        """One update on the gathered slots: `ident` maps (update cursor, batch item) to a row of X / G / Y; `slot` is the
        slot this update of the captured sequence reads (only the 'rccl' form needs it spelled out)."""
        from .. import _lib, ops
        L, o = _lib.lib(), self.learner.actor_optim
        if self.aggregated:
            return self._train_agg(ident, slot)
        if self.dp == 'p2p':
            _lib.check(L.mgp_train_step_p2p(
                ops._ptr(self.X), ops._ptr(self.G), ops._ptr(self.Y), ident.data_ptr(), self.cursor.data_ptr(),
                ops._ptr(self.loss_hist), self.cap, ops._ptr(o.flat), ops._ptr(o.flat_grad), ops._ptr(o.m), ops._ptr(o.v),
                self.cdims, self.nl, o.lr, o.betas[0], o.betas[1], o.eps, self.step_dev.data_ptr(), None, ops._ptr(self.ws),
                self.B, self.K, self.N, self.learner.p2p.handle, ops._stream()), 'mgp_train_step_p2p')
            return
        if self.dp == 'rccl':
            # gradients of this rank's minibatch (+ its loss) into one flat buffer, ONE all-reduce, the step with the
            # round's bookkeeping (loss filed at the cursor, cursor advanced) -- all captured
            import torch.distributed as dist
            B, P = self.B, o.flat.numel()
            lo = slot * B
            _lib.check(L.mgp_train_grads(ops._ptr(self.X[lo:lo + B]), ops._ptr(self.G[lo:lo + B]), ops._ptr(self.Y[lo:lo + B]),
                                         self.Wp, self.bp, self.cdims, self.nl, ops._ptr(self.xbuf), ops._ptr(self.xbuf[P:]),
                                         ops._ptr(self.ws), B, self.K, self.N, ops._stream()), 'mgp_train_grads')
            dist.all_reduce(self.xbuf, op=dist.ReduceOp.SUM)
            self.xbuf.div_(parallel.world_size())
            _lib.check(L.mgp_adam_step_filed(ops._ptr(o.flat), ops._ptr(self.xbuf), ops._ptr(o.m), ops._ptr(o.v), P, o.lr,
                                             o.betas[0], o.betas[1], o.eps, self.step_dev.data_ptr(), ops._ptr(self.xbuf[P:]),
                                             ops._ptr(self.loss_hist), self.cap, self.cursor.data_ptr(), ops._stream()),
                       'mgp_adam_step_filed')
            return
        _lib.check(L.mgp_train_step_indexed(
            ops._ptr(self.X), ops._ptr(self.G), ops._ptr(self.Y), ident.data_ptr(), self.cursor.data_ptr(),
            ops._ptr(self.loss_hist), self.cap, ops._ptr(o.flat), ops._ptr(o.flat_grad), ops._ptr(o.m), ops._ptr(o.v),
            self.cdims, self.nl, o.lr, o.betas[0], o.betas[1], o.eps, self.step_dev.data_ptr(), ops._ptr(self.ws),
            self.B, self.K, self.N, ops._stream()), 'mgp_train_step_indexed')

    def _enqueue(self):
        """one update: gather into slot 0, train on it"""
        from .. import ops
        B = self.B
        if self.aggregated:
            ops.replay_aggregate(self.memory, self.idx, self.Z[:B], self.Y[:B], self.mean_pooling, cursor=self.cursor)
        else:
            ops.replay_gather(self.memory, self.idx, self.X[:B], self.G[:B], self.Y[:B], self.mean_pooling, cursor=self.cursor)
        self._train(self.ident)

    def _enqueue_many(self, n):
        """n updates starting at a cursor that is a multiple of n: one gather for all of them, then the n train steps"""
        from .. import ops
        assert n == UPDATES_PER_GRAPH
        if self.aggregated:
            ops.replay_aggregate(self.memory, self.idx, self.Z, self.Y, self.mean_pooling, cursor=self.cursor, nb=n)
        else:
            ops.replay_gather(self.memory, self.idx, self.X, self.G, self.Y, self.mean_pooling, cursor=self.cursor, nb=n)
        for i in range(n):
            self._train(self.ident_many, slot=i)

    def run_sampled(self, U, sampler=None):
        """U updates, frame indices drawn per update by `sampler` (default: FrameReplay.sample_ids -- the reference's
        random.sample over the buffer positions), host sampling overlapped with the GPU (see _run_sampled)."""
        m = self.memory
        if sampler is None:
            return _run_sampled(self, U, None, batch_sampler=lambda n_: m.sample_ids_many(self.B, n_))
        return _run_sampled(self, U, sampler)

    def run(self, ids):
        """ids: one list of `batch_size` frame indices per update.  Returns the sum of the updates' losses (device tensor)."""
        U = len(ids)
        assert 0 < U <= self.cap
        self.idx[:U].copy_(torch.tensor(ids, dtype=torch.long), non_blocking=False)
        self.cursor.zero_()
        _replay_updates(self, U)
        self.learner.actor_optim.step_count += U
        return self.loss_hist[:U].sum()


def _params_from_args(args):
    env_cls = _REGISTRY.get(args.get('env'), None)
    variant = getattr(env_cls, 'variant', {}) if env_cls is not None else {}
    kw = dict(n_agents=args.getint('n_agents'), comm_radius=args.getfloat('comm_radius'),
              v_max=args.getfloat('v_max'), v_bias=args.getfloat('v_max'))
    # same reset distribution as the gym-style environment built from the same cfg section (FlockParams.init_mode 'auto':
    # disc sampling up to N = 100, jittered lattice beyond), so `alg = dagger_vec` and `alg = dagger` statistics are
    # comparable; `init_mode = grid` in the cfg selects the lattice explicitly (cheaper resets for many lanes)
    if args.get('init_mode') is not None:
        kw['init_mode'] = args.get('init_mode')
    if args.get('dt') is not None:
        kw['dt'] = args.getfloat('dt')
    kw.update(variant)
    if args.get('link_drop') is not None:
        kw['link_drop'] = args.getfloat('link_drop')
    if kw.get('link_drop', 0.0) > 0.0:
        kw['link_seed'] = args.getint('seed', fallback=0) & 0xFFFFFFFF
    return FlockParams(**kw)


def _label(expert):
    """(B,N,nA) -> (B,1,nA,N)   (reference gnn_dagger.py:174-176, batched)"""
    return expert.permute(0, 2, 1).unsqueeze(1).contiguous()


def evaluate(learner, sim, state, n_episodes, steps):
    """Policy-only episodes in lanes of sim.B; returns the list of per-episode reward sums."""
    rewards = []
    while len(rewards) < n_episodes:
        sim.reset(np.random)
        state.reset()
        state.push(sim.network, sim.features)
        # whole episodes in one launch of the episode-resident kernel when the shape is covered (else step by step)
        per_step = torch.zeros((sim.B, steps), device=sim.device, dtype=torch.float64)
        policy_rollout(learner.actor, sim, state, steps, rewards=per_step)
        total = per_step.sum(dim=1)
        rewards += total.cpu().tolist()
    return rewards[:n_episodes]


def collect_supported(learner, K, N, params=None):
    """The collecting builds of the episode-resident kernels cover the shape (mgp_rollout_collect: N <= 256, 6 features,
    2-D actions, aggregation in front of the first layer, widths <= 64 or one hidden layer <= 128) AND the simulator's networks have symmetric
    membership (`params.symmetric_network`): the frame replay stores bit ROWS and rebuilds `A_t A_{t-1} ...` reading row n as
    column n, which a directed network would silently get wrong."""
    from .. import ops
    if params is not None and not getattr(params, 'symmetric_network', True):
        return False
    if learner.actor.ind_agg != 0 or learner.n_states != 6:
        return False
    if N > 256:                                                 # factored state in HBM: mgp_sparse_policy_collect, K launches per step
        from .sparse_rollout import sparse_supported
        return sparse_supported(learner.actor, K, N)
    hidden = [int(v) for v in learner.actor.layers[1:-1]]
    if len(hidden) >= 2 and max(hidden) > 64:                   # two 128-wide layers: the resident build that streams its second layer
        return False                                            # (rollout_w128x2.hip) has no collecting form
    return ops.rollout_supported(tuple(learner.actor.layers), K, N)


def collect_round(learner, sim, state, memory, beta, episode_ids, seed, T, chunk=None, x0=None):
    """One lock-step round of DAGGER data collection ON THE DEVICE (reference gnn_dagger.py:150-178 for every lane): reset,
    then T steps inside mgp_rollout_collect launches -- policy forward, expert label, beta coin, simulator step, state
    transition and the filing of every visited state into the frame ring all happen in the kernel; the host draws the reset
    states (or is handed them: `x0` (n_envs, N, 4), drawn ahead from the same generator) and launches.  `beta` (n_envs,)
    float32 and `episode_ids` (n_envs,) int32 on the device."""
    from .. import ops
    from .rollouts import _actor_params
    if x0 is None:
        sim.reset(np.random)
    else:
        sim.set_state(x0)                                       # reset states drawn ahead (train_dagger_vec): the same draws, earlier
    state.reset()
    state.push(sim.network, sim.features)                      # reset observation: all-zero operator history (carry)
    if sim.N > 256:
        # beyond the LDS-resident kernel: the same round on the factored state in HBM -- K launches per env step, the frame,
        # the coin and the label inside the policy launch (mgp_sparse_policy_collect); still no host RNG, no per-step traffic
        from .sparse_rollout import SparseFlockState, sparse_collect
        sp = SparseFlockState(sim, state.K)
        sp.observe_reset(sim)
        sparse_collect(learner.actor, sim, sp, memory, beta, episode_ids, seed, 0, T)
        sp.to_dense(sim, state, lazy=True)
        state._pushes += T
        sp.at_push = state._pushes
        state._sparse = sp
        return
    expert_io = sim.controller().permute(0, 2, 1).contiguous() # (B,2,N): the expert's action for the reset state
    ws, bs = _actor_params(learner.actor)
    image = ops.rollout_image(ws, bs, tuple(learner.actor.layers), state.K, sim.N)      # weights are fixed for the round
    carry = state.carry_buffer()
    assert carry is not None and state._carry_valid
    flags = ops.RO_ENTER_CARRY | ops.RO_EXIT_CARRY | ops.RO_SKIP_DENSE
    done = 0
    while done < T:
        t = min(chunk or T, T - done)
        ok = ops.rollout_collect(sim.x, state._G[state._cur], state.delay_state, tuple(learner.actor.layers), sim._c, t,
                                 memory, expert_io, beta, episode_ids, seed, age0=done, ring_step0=memory.head, carry=carry,
                                 flags=flags, image=image)
        assert ok
        memory.advance(t)
        state._pushes += t
        state._dense_stale = True
        done += t
    sim._network, sim._network_lazy = None, (lambda: state.delay_gso[:, 1])
    sim.features = state.delay_state[:, 0]


def train_dagger_vec(args, device, n_envs=64, episode_steps=None):
    """Returns {'mean','std'} like train_dagger.  `n_envs` parallel episodes per rank.

    Where the collecting builds of the episode-resident kernels cover the shape (N <= 256: every N of the reference's
    sweeps) a round is ONE launch per GPU for the rollouts of all lanes -- no host RNG, no host<->device traffic
    per step -- into a compact frame replay, and one HIP-graph replay per update.  Larger flocks collect on the factored
    state in HBM (K launches per env step, frame / coin / label inside the policy launch; frames of 160 KB at N = 1000 where
    the dense pair is 12 MB).  Other shapes step the two-launch path from the host and keep dense states in the replay
    (the round-1 loop)."""
    device = torch.device(device)
    p = _params_from_args(args)
    N, K, F, n_a = p.n_agents, args.getint('k'), args.getint('n_states'), args.getint('n_actions')
    T = episode_steps or p.max_episode_steps
    learner = DAGGER(device, args)
    on_device = collect_supported(learner, K, N, p) and args.get('collect', 'device') != 'host'
    # The reference's ring of `buffer_size` (10,000) transitions holds its 20 most recent WHOLE episodes.  Here n_envs
    # episodes advance in lock step, so a ring shorter than one round (n_envs * T transitions) would keep only the last
    # steps of every episode -- the already-flocked states -- and the policy would never see a start-up state.  The ring
    # therefore holds at least one full round (4.8 KB per transition as frames at N = 100: 154 MB for 64 x 500; the dense
    # fallback takes 128 KB per transition, 4 GB).
    capacity = max(args.getint('buffer_size'), n_envs * T)
    memory = (FrameReplay(n_envs, capacity, K, N, device) if on_device
              else DeviceReplay(capacity, K, F, N, n_a, device))
    sim = VecFlock(n_envs, p, device, with_expert=True)
    state = BatchedDelayState(device, n_envs, K, F, N)
    batch_size = args.getint('batch_size')
    beta_of = BetaSchedule(args.getfloat('beta_coeff'))
    updates_per_step = args.getint('updates_per_step')
    n_train_episodes = args.getint('n_train_episodes')
    n_test_episodes = args.getint('n_test_episodes')
    seed = args.getint('seed', fallback=0)
    debug = args.getboolean('debug')
    rank, world = parallel.rank(), parallel.world_size()
    rounds = (n_train_episodes + n_envs * world - 1) // (n_envs * world)
    updates = 0
    indexed = None
    next_x0, side = None, None
    import os
    prefetch_resets = os.environ.get('MGP_PREFETCH_RESETS', '1') != '0'        # (0: every round draws its resets when it starts)
    for rd in range(rounds):
        e0 = (rd * world + rank) * n_envs
        beta = np.array([beta_of(e) for e in range(e0, e0 + n_envs)], dtype=np.float64)   # reference schedule per global episode
        if on_device:
            collect_round(learner, sim, state, memory, torch.tensor(beta, dtype=torch.float32, device=device),
                          torch.arange(e0, e0 + n_envs, dtype=torch.int32, device=device), seed, T, x0=next_x0)
            next_x0 = None
        else:
            sim.reset(np.random)
            state.reset()
            state.push(sim.network, sim.features)
            for _ in range(T):
                expert = sim.controller()                                     # by-product of the last sim kernel
                memory.insert_batch(state.delay_state, state.delay_gso, _label(expert))
                with torch.no_grad():
                    policy = learner.actor(state.delay_state, state.delay_gso)   # (B,1,nA,N)
                use_expert = torch.from_numpy(np.random.binomial(1, beta).astype(np.bool_)).to(device)
                action = torch.where(use_expert.view(-1, 1, 1), expert, policy[:, 0].permute(0, 2, 1)).contiguous()
                A_dst, X_dst = state.next_slots()
                sim.step(action, A_out=A_dst, feat_out=X_dst)
                state.advance()
        loss_sum = 0.0
        n_updates = updates_per_step * n_envs
        if n_updates > 0 and memory.curr_size > batch_size and (FrameUpdates if on_device else IndexedUpdates).supported(
                learner, batch_size, N):
            if indexed is None:
                indexed = (FrameUpdates(learner, memory, batch_size, n_updates, p.mean_pooling) if on_device
                           else IndexedUpdates(learner, memory, batch_size, n_updates))
            learner.begin_updates()                                     # data parallel: ranks aligned before the exchanges
            loss_dev = indexed.run_sampled(n_updates)                   # random.sample per update; enqueued, not waited for
            if on_device and rd + 1 < rounds and not use_grid(p) and prefetch_resets:
                # the next round's reset states, drawn while the GPU runs this round's updates: the rejection sampler costs
                # ~1 ms of host time per disc reset (MT19937 + libm for ~140 candidates), nothing between here and the next
                # collect_round draws from numpy's generator, and the acceptance launches go to a stream of their own
                if side is None:
                    side = torch.cuda.Stream(device)
                with torch.cuda.stream(side):
                    next_x0 = sample_initial_states(np.random, p, n_envs, device)
            loss_sum = float(loss_dev.item())
            learner.end_updates()
            updates += n_updates
        elif n_updates > 0 and memory.curr_size > batch_size:
            bufs = learner.graphed_buffers(batch_size, N)        # None: composed eager updates (shape outside the fused kernels)
            graphed = bufs is not None
            if bufs is None and on_device:
                bufs = tuple(torch.empty(sh, device=device) for sh in ((batch_size, K, F, N), (batch_size, K, N, N),
                                                                       (batch_size, 1, n_a, N)))
            loss_dev = torch.zeros((1,), device=device)
            learner.begin_updates()                               # as above: aligned ranks, then the status of the exchanges
            for _ in range(n_updates):
                if on_device:
                    xs, gs, ys = memory.sample(batch_size, bufs, mean_pooling=p.mean_pooling)
                else:
                    xs, gs, ys = memory.sample(batch_size, out=bufs)
                if graphed:                                       # no host sync per update: losses add up on the device
                    loss_dev += learner.gradient_step_tensors(xs, gs, ys, sync=False)
                else:
                    loss_sum += learner.gradient_step_tensors(xs, gs, ys)
                updates += 1
            loss_sum += float(loss_dev.item())
            learner.end_updates()
        if debug and rank == 0:
            print("Round: {}, episodes: {}, updates: {}, policy loss: {}".format(rd, (rd + 1) * n_envs * world,
                                                                                   updates, loss_sum))
    lo, hi = parallel.shard_range(n_test_episodes)
    n_local = max(1, hi - lo) if world > 1 else n_test_episodes
    rewards = parallel.all_gather_floats(evaluate(learner, sim, state, n_local, T))
    if debug and args.get('fname') and rank == 0:            # reference gnn_dagger.py:239-240
        learner.save_model(args.get('env'), suffix=args.get('fname'))
    return {'mean': float(np.mean(rewards)), 'std': float(np.std(rewards)), 'learner': learner, 'updates': updates,
            'collect': 'device' if on_device else 'host', 'replay_bytes_per_transition':
            memory.bytes_per_transition() if on_device else 4 * (K * F * N + K * N * N + n_a * N)}
