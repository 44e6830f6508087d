"""`MultiAgentStateWithDelay` -- drop-in for reference learner/state_with_delay.py:4-53.

Same constructor `(device, args, env_state, prev_state=None, k=None)` and attributes
`values (1,1,F,N)`, `network (1,1,N,N)`, `delay_gso (1,K,N,N)`, `delay_state (1,K,F,N)`,
`curr_gso (1,K,N,N)`.  The recursion `delay_gso[1:] = A_t @ prev.delay_gso[:-1]` and the delay line
run in the HIP kernel `mgp_gso_update`.  `curr_gso` (powers of A_t) is only consumed by the reference's
dead DDPG path, so it is computed lazily on first access (`mgp_gso_powers`).

`BatchedDelayState` is the B-episode, allocation-free form used by the vectorised rollout.
"""
import numpy as np
import torch

from .. import ops


class MultiAgentStateWithDelay(object):

    def __init__(self, device, args, env_state, prev_state=None, k=None):
        # the three cfg lookups of the reference constructor (state_with_delay.py:14-16), parsed once per cfg section:
        # configparser's getint costs ~6 us a call and this constructor runs once per environment step
        shape = getattr(args, '_mgp_state_shape', None)
        if shape is None:
            shape = (args.getint('n_states'), args.getint('n_agents'), args.getint('k'))
            try:
                args._mgp_state_shape = shape
            except AttributeError:
                pass
        n_states, n_agents = shape[0], shape[1]
        k = k or shape[2]

        state_value, state_network = env_state
        # contract of reference state_with_delay.py:24-26
        assert state_value.shape == (n_agents, n_states)
        assert state_network.shape == (n_agents, n_agents)
        on_device = hasattr(state_value, 'device32') and hasattr(state_network, 'device32')
        if on_device:
            assert state_network.zero_diagonal      # guaranteed by the simulator kernel (no self loops)
        else:
            assert np.sum(np.diag(state_network)) == 0  # no self loops

        device = torch.device(device)
        if device.type != 'cuda':
            raise ops.MgpError("MultiAgentStateWithDelay needs a HIP device (got %s); this package has no CPU "
                               "compute path" % device)
        if on_device:
            # observation produced by this package's simulator: the fp32 tensors are already on the GPU
            self.values = state_value.device32.to(device).reshape(1, 1, n_states, n_agents)
            self.network = state_network.device32.to(device).reshape(1, 1, n_agents, n_agents)
        else:
            # fp64 -> fp32 on the host (what torch.Tensor(ndarray) does, state_with_delay.py:34-35), then H2D
            v32 = np.ascontiguousarray(np.asarray(state_value).T, dtype=np.float32).reshape(1, 1, n_states, n_agents)
            a32 = np.ascontiguousarray(np.asarray(state_network), dtype=np.float32).reshape(1, 1, n_agents, n_agents)
            self.values = torch.from_numpy(v32).to(device)
            self.network = torch.from_numpy(a32).to(device)
        self._k = k

        has_prev = prev_state is not None and k > 1
        G_prev = prev_state.delay_gso if has_prev else None
        Xd_prev = prev_state.delay_state if has_prev else None
        self.delay_gso, self.delay_state = ops.gso_update(self.network[0], G_prev, self.values[0], Xd_prev, k)
        self._curr_gso = None

    @property
    def curr_gso(self):
        """I, A_t, A_t^2, ... (reference state_with_delay.py:38-41); lazily evaluated."""
        if self._curr_gso is None:
            self._curr_gso = ops.gso_powers(self.network[0], self._k)
        return self._curr_gso


class BatchedDelayState(object):
    """Delay line + delayed GSO for B independent episodes, resident on the device.

    Holds two (B,K,N,N) / (B,K,F,N) ping-pong buffer pairs; `push(A, X)` advances every episode by one
    step with a single `mgp_gso_update` launch and no allocation (HIP-graph capturable).
    """

    def __init__(self, device, B, K, F, N):
        self.B, self.K, self.F, self.N = B, K, F, N
        kw = dict(device=device, dtype=torch.float32)
        self._G = [torch.zeros((B, K, N, N), **kw), torch.zeros((B, K, N, N), **kw)]
        self._X = [torch.zeros((B, K, F, N), **kw), torch.zeros((B, K, F, N), **kw)]
        for g in self._G:
            g[:, 0] = torch.eye(N, **kw)          # slice 0 is the identity for the whole life of the buffer
        self._scratch_A = torch.zeros((B, N, N), **kw) if K == 1 else None
        self._cur = 0
        self._has_prev = False
        self._pushes = 0                          # states pushed since the last reset (1 = the reset observation only)
        # Factored hand-over between launches of the episode-resident kernel (mgp_rollout_steps_ex): `_carry` holds the
        # membership bits + row weights of the last K-1 networks of the CURRENT state while `_carry_valid`; the dense slices
        # delay_gso[:, 1:] of the current buffer are materialised from it on first use while `_dense_stale`
        # (mgp_rollout_carry_to_dense: same arithmetic as the in-launch rebuild).  Any transition made outside the
        # resident kernel invalidates the carry; a reset observation has the all-zero carry (no earlier network).
        self._carry = None
        self._carry_valid = False
        self._dense_stale = False
        self._dense_from = None                   # factored state in HBM (N > 256): callable that rebuilds the dense slices

    def reset(self):
        self._has_prev = False
        self._pushes = 0
        self._carry_valid = False
        self._dense_stale = False
        self._dense_from = None

    def carry_buffer(self):
        """(B, mgp_rollout_carry_bytes) uint8 device buffer, or None when the resident kernel does not cover (K, N)."""
        if self._carry is None:
            nbytes = ops.rollout_carry_bytes(self.K, self.N)
            if nbytes <= 0:
                return None
            self._carry = torch.zeros((self.B, nbytes), device=self._G[0].device, dtype=torch.uint8)
        return self._carry

    def _ensure_dense(self):
        if self._dense_stale:
            self._dense_stale = False
            if self._dense_from is not None:      # left by a rollout on the factored state (sparse_rollout.py)
                fn, self._dense_from = self._dense_from, None
                fn()
            else:
                ops.rollout_carry_to_dense(self._carry, self._G[self._cur], self.K)

    def push(self, A, X_t):
        """A (B,N,N) fp32, X_t (B,F,N) fp32 on the device.  mgp_gso_update reads both with batch strides N*N and F*N, so
        strided views (e.g. the slots of another delay state that a simulator step wrote into) are compacted first."""
        assert A.shape == (self.B, self.N, self.N) and X_t.shape == (self.B, self.F, self.N)
        if not A.is_contiguous():
            A = A.contiguous()
        if not X_t.is_contiguous():
            X_t = X_t.contiguous()
        self._ensure_dense()
        nxt = 1 - self._cur
        ops.gso_update_into(A, self._G[self._cur], self._G[nxt], X_t, self._X[self._cur], self._X[nxt],
                            has_prev=self._has_prev)
        self._cur = nxt
        # the reset observation has no history: its carry is all zeros (every delayed product vanishes, as the zero-filled
        # slices of the reference do, state_with_delay.py:44-47); later pushes build on dense slices only
        self._carry_valid = False
        if not self._has_prev and self.carry_buffer() is not None:
            self._carry.zero_()
            self._carry_valid = True
        self._has_prev = True
        self._pushes += 1

    # ---- in-place protocol: the simulator writes A_t / X_t straight into the next buffers ---------------
    def next_slots(self):
        """(A_dst (B,N,N) view = next delay_gso[:,1], X_dst (B,F,N) view = next delay_state[:,0]): hand these to
        VecFlock.step(..., A_out=, feat_out=) and then call advance().  Saves the A read-copy-write and the identity
        rewrite of push() (3 N^2 instead of 5 N^2 floats of state traffic per episode-step at K = 3)."""
        self._ensure_dense()
        nxt = 1 - self._cur
        A_dst = self._G[nxt][:, 1] if self.K > 1 else self._scratch_A
        return A_dst, self._X[nxt][:, 0]

    def buffers(self):
        """(G_prev, G_next, Xd_prev, Xd_next) for a fused sim+state kernel; call flip() after it ran."""
        self._ensure_dense()
        nxt = 1 - self._cur
        return self._G[self._cur], self._G[nxt], self._X[self._cur], self._X[nxt]

    @property
    def has_prev(self):
        return self._has_prev

    def flip(self):
        self._cur = 1 - self._cur
        self._has_prev = True
        self._pushes += 1
        self._carry_valid = False

    def advance(self):
        self._ensure_dense()
        self._carry_valid = False
        nxt = 1 - self._cur
        ops.gso_advance(self._G[self._cur], self._G[nxt], self._X[self._cur], self._X[nxt], has_prev=self._has_prev)
        self._cur = nxt
        self._has_prev = True
        self._pushes += 1

    @property
    def delay_gso(self):
        self._ensure_dense()
        return self._G[self._cur]

    @property
    def delay_state(self):
        return self._X[self._cur]
