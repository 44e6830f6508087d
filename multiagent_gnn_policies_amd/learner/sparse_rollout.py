"""Closed-loop policy rollouts on the factored episode state, for flocks beyond the LDS-resident kernel (N > 256).

Same loop as `policy_rollout` (reference test_model.py:38-44) and the same mathematics as `mgp_rollout_steps`: tap j of the
Actor's aggregation is x_{t-j} A_t A_{t-1} ... A_{t-j+1} (state_with_delay.py:44-47, actor.py:64-71), evaluated left to right
along the networks' membership bit rows.  The state is three ring buffers in HBM -- bit rows and row weights of the last
K - 1 networks, the last K feature blocks as (N, 8) rows -- and one environment step is K kernel launches (K - 2 gather
stages, the policy tail, the simulator).  The dense `delay_gso (B,K,N,N)` of the reference contract is produced on demand
(`SparseFlockState.to_dense`), not every step: at N = 1000 it is 12 MB per episode.

A factored state can only be started where the history is known: at a reset (no history: missing networks are zero
rows with zero weights, missing features zeros, which makes the products vanish exactly as the reference's zero-filled
state does, state_with_delay.py:47-53) or carried over from the previous sparse rollout of the same simulator.
"""
import ctypes

import torch

from .. import _lib, ops
from .actor_fused import _ptr_array


import os
_NOLISTS = os.environ.get('MGP_SP_NOLISTS', '0') == '1'        # debugging aid: keep the gather launches on the bit rows


class SparseFlockState(object):
    """Bit rows / row weights / features of B episodes of `sim` (VecFlock), ring-indexed over time."""

    def __init__(self, sim, K):
        self.B, self.N, self.K = sim.B, sim.N, K
        self.H = K - 1 if K > 2 else 1
        self.NW = _lib.lib().mgp_sparse_words(self.N)
        dev = sim.device
        self.bits = torch.zeros((self.B, self.H, self.N, self.NW), device=dev, dtype=torch.int64)
        self.wrow = torch.zeros((self.B, self.H, self.N), device=dev, dtype=torch.float32)
        self.feat = torch.zeros((self.B, K, self.N, 8), device=dev, dtype=torch.float32)
        self.scratch = torch.zeros((max(1, 2 * (K - 1) * self.B * self.N * 8),), device=dev, dtype=torch.float32)
        # the same networks once more as compact neighbour lists (mgp_flock_step_cells_nbr; the gather launches read these
        # instead of the bit rows).  Valid while every network of the ring was written by the cell-list simulator.
        self.nbr = (torch.zeros((self.B, self.H, self.N, 16), device=dev, dtype=torch.int16)
                    if (self.N <= 2048 and not _NOLISTS) else None)
        self._nbr_ok = False
        self.cur = 0            # ring slot of x_t in feat
        self.hs = 0             # ring slot of A_t in bits / wrow
        self.steps = 0          # simulator steps since the reset this state was started at
        self.owner = None       # the tensor sim.x pointed at when this state was last advanced (continuity check)
        self.use_cells = True

    def _sim_call(self, sim, x_in, x_out, u, h, c, reward, expert):
        L = _lib.lib()
        su_agent, su_axis = (1, self.N) if u is not None else (2, 1)          # the Actor's output layout (B,1,2,N)
        # cell-list simulator up to N = 2048 (identical bit rows), the all-pairs kernel beyond
        if self.N <= 2048 and self.use_cells:
            rc = L.mgp_flock_step_cells_nbr(
                ops._ptr(x_in), ops._ptr(x_out), ops._ptr(u), su_agent, su_axis,
                self.bits.data_ptr() + h * self.N * self.NW * 8, self.H * self.N * self.NW,
                self.wrow.data_ptr() + h * self.N * 4, self.H * self.N,
                self.feat.data_ptr() + c * self.N * 8 * 4, self.K * self.N * 8,
                (self.nbr.data_ptr() + h * self.N * 16 * 2) if self.nbr is not None else None, self.H * self.N * 16,
                ops._ptr(reward), ops._ptr(expert), ctypes.byref(sim._c), self.B, self.N, ops._stream())
            _lib.check(rc, 'mgp_flock_step_cells_nbr')
            return
        self._nbr_ok = False                                     # the all-pairs kernel writes bit rows only
        rc = L.mgp_flock_step_sparse(
            ops._ptr(x_in), ops._ptr(x_out), ops._ptr(u), su_agent, su_axis,
            self.bits.data_ptr() + h * self.N * self.NW * 8, self.H * self.N * self.NW,
            self.wrow.data_ptr() + h * self.N * 4, self.H * self.N,
            self.feat.data_ptr() + c * self.N * 8 * 4, self.K * self.N * 8,
            ops._ptr(reward), ops._ptr(expert), ctypes.byref(sim._c), self.B, self.N, ops._stream())
        _lib.check(rc, 'mgp_flock_step_sparse')

    def check_status(self):
        """Synchronises; raises if an episode's persistent workgroups gave up waiting for each other in an earlier
        mgp_sparse_rollout call on this state (csrc/sparse_persist.hip: CUs held by another process; its outputs are NaN)."""
        rc = _lib.lib().mgp_sparse_rollout_status(ops._ptr(self.scratch), self.B, self.K, self.N, ops._stream())
        _lib.check(rc, 'mgp_sparse_rollout_status')

    def observe_reset(self, sim):
        """Start at the simulator's current x as a freshly reset episode (no history)."""
        self.bits.zero_(); self.wrow.zero_(); self.feat.zero_()
        if self.nbr is not None:
            self.nbr.zero_()                                     # count 0: a network that does not exist yet has no neighbours
            self._nbr_ok = self.use_cells
        self.cur = self.hs = self.steps = 0
        self._sim_call(sim, sim.x, sim._x_next, None, 0, 0, None, sim.expert if sim.with_expert else None)
        self.owner = sim.x

    def step(self, sim, action):
        """Simulator step with the policy output `action` (B,1,2,N); the new network / features enter the rings."""
        nh = (self.hs + 1) % self.H
        nc = (self.cur + 1) % self.K
        self._sim_call(sim, sim.x, sim._x_next, action, nh, nc, sim.reward, sim.expert if sim.with_expert else None)
        sim.x, sim._x_next = sim._x_next, sim.x
        self.hs, self.cur = nh, nc
        self.steps += 1
        self.owner = sim.x

    def to_dense(self, sim, state, lazy=False):
        """Materialise the reference's dense state into `state` (BatchedDelayState): delay_state from the feature ring now,
        delay_gso slices 1..K-1 from the bit rows now or -- lazy -- when somebody reads them (state.delay_gso /
        sim.network: 12 MB per episode at N = 1000, 180 us for 64 episodes), and point the simulator's observation views at them."""
        self.check_status()
        X = state._X[state._cur]
        for k in range(self.K):                                 # (K small strided copies; no index tensor: that would be an H2D)
            X[:, k].copy_(self.feat[:, (self.cur - k) % self.K, :, :6].transpose(1, 2))
        hs, bits, wrow = self.hs, self.bits, self.wrow

        def build():
            rc = _lib.lib().mgp_sparse_to_dense(bits.data_ptr(), ops._ptr(wrow), ops._ptr(state._G[state._cur]), self.B,
                                                self.K, self.N, hs, ops._stream())
            _lib.check(rc, 'mgp_sparse_to_dense')
        state._has_prev = True
        state._carry_valid = False
        if lazy:
            # the rings move on with the next step: the closure is only valid until then, which is exactly as long as
            # `state` describes this state (any later transition goes through _ensure_dense first)
            state._dense_stale, state._dense_from = True, build
        else:
            state._dense_stale, state._dense_from = False, None
            build()
        if self.K > 1:
            sim._network, sim._network_lazy = None, (lambda: state.delay_gso[:, 1])
        sim.features = X[:, 0]


def sparse_supported(actor, K, N):
    dims = tuple(actor.layers)
    cd = (ctypes.c_int * len(dims))(*dims)
    return actor.ind_agg == 0 and bool(_lib.lib().mgp_sparse_policy_supported(cd, len(dims) - 1, K, N))


def _policy_image(actor, sim, K):
    """(dims, n_layers, weight image) of `actor` for the factored policy kernels.  Rebuilt on every call: the optimiser's
    kernels write the parameters through raw pointers, so torch's version counters do not see a training update."""
    L = _lib.lib()
    dims = tuple(actor.layers)
    cd = (ctypes.c_int * len(dims))(*dims)
    nl = len(dims) - 1
    Ws = [c.weight.detach().reshape(c.weight.shape[0], -1).contiguous() for c in actor.conv_layers]
    bs = [c.bias.detach().contiguous() for c in actor.conv_layers]
    image = torch.empty((L.mgp_sparse_policy_image_floats(cd, nl, K),), device=sim.device, dtype=torch.float32)
    _lib.check(L.mgp_sparse_policy_image(_ptr_array(Ws), _ptr_array(bs), cd, nl, K, ops._ptr(image), ops._stream()),
               'mgp_sparse_policy_image')
    return cd, nl, image


def _run_steps(sim, sp, cd, nl, image, act, T, rw, collect):
    """T steps enqueued by one library call (mgp_sparse_rollout); keeps `sp` / `sim` in step with the rings and the ping-pong."""
    cur, hs = ctypes.c_int(sp.cur), ctypes.c_int(sp.hs)
    assert sim.with_expert or collect is None
    rc = _lib.lib().mgp_sparse_rollout(
        sp.bits.data_ptr(), ops._ptr(sp.wrow), ops._ptr(sp.feat), ops._ptr(image), cd, nl, ops._ptr(sp.scratch), ops._ptr(act),
        ops._ptr(sim.x), ops._ptr(sim._x_next), ops._ptr(rw), ops._ptr(sim.expert if sim.with_expert else None),
        ctypes.byref(sim._c), sp.B, sp.K, sp.N, int(T), ctypes.byref(cur), ctypes.byref(hs),
        ctypes.byref(collect) if collect is not None else None,
        sp.nbr.data_ptr() if (sp.nbr is not None and sp._nbr_ok and not _NOLISTS) else None, ops._stream())
    _lib.check(rc, 'mgp_sparse_rollout')
    if T & 1:
        sim.x, sim._x_next = sim._x_next, sim.x
    sp.cur, sp.hs = cur.value, hs.value
    sp.steps += T
    sp.owner = sim.x


def sparse_collect(actor, sim, sp, frames, beta, episode_ids, seed, age0, T, rewards=None):
    """T DAGGER data-collection steps on the factored state (reference gnn_dagger.py:154-178 for every lane; the semantics
    of mgp_rollout_collect for N > 256): every step files the state it starts from into `frames` (FrameReplay with .wrow)
    at ring step frames.head, and is driven by the expert -- `sim.expert`, the by-product of the simulator kernel that
    produced the current state -- where the lane's counter-based coin says so, else by the policy.  K launches per env step;
    the weights are fixed for the call.  `sim` must be built with_expert; advances frames.head."""
    L = _lib.lib()
    K, B, N = sp.K, sp.B, sp.N
    assert sim.with_expert and frames.wrow is not None and frames.lanes == B
    assert beta.shape == (B,) and beta.dtype == torch.float32 and episode_ids.shape == (B,) and episode_ids.dtype == torch.int32
    cd, nl, image = _policy_image(actor, sim, K)
    act = torch.empty((B, 1, 2, N), device=sim.device, dtype=torch.float32)
    rw = torch.empty((T, B), device=sim.device, dtype=torch.float64) if rewards is not None else None
    cl = _lib.MgpSparseCollect(frames.feat.data_ptr(), frames.bits.data_ptr(), frames.wrow.data_ptr(),
                               frames.label.data_ptr(), frames.age.data_ptr(), sim.expert.data_ptr(), beta.data_ptr(),
                               episode_ids.data_ptr(), int(seed) & 0xFFFFFFFF, int(age0), frames.head, frames.ring_steps)
    _run_steps(sim, sp, cd, nl, image, act, T, rw, cl)
    frames.advance(T)
    if rw is not None:
        rewards.copy_(rw.t())
        sim.reward.copy_(rw[T - 1])
    return True


def sparse_policy_rollout(actor, sim, sp, T, rewards=None, action=None):
    """T closed-loop policy steps on the factored state `sp` (SparseFlockState of `sim`).  rewards (B,T) fp64 and
    action (B,1,2,N) as in `policy_rollout`.  K launches per step, nothing else on the device."""
    L = _lib.lib()
    K, B, N = sp.K, sp.B, sp.N
    cd, nl, image = _policy_image(actor, sim, K)
    act = action if action is not None else torch.empty((B, 1, 2, N), device=sim.device, dtype=torch.float32)
    rw = torch.empty((T, B), device=sim.device, dtype=torch.float64) if rewards is not None else None
    _run_steps(sim, sp, cd, nl, image, act, T, rw, None)
    if rw is not None:
        rewards.copy_(rw.t())
        sim.reward.copy_(rw[T - 1])
    return True
