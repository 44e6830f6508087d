"""Flocking environments with the call-site contract the reference expects from `gym_flock`
(reference train.py:18-25, gnn_dagger.py:150-163, gnn_baseline.py:13-17, state_with_delay.py:22-26):

    env = make('FlockingRelative-v0'); env.env.params_from_cfg(args); env.seed(s)
    (values (N,6) f64, network (N,N) f64) = env.reset()
    u = env.env.controller()                      # (N,2)
    (values, network), reward, done, info = env.step(u)

`gym` / `gym_flock` are not dependencies (neither is installable here); `make` and `TimeLimit` provide
the two things the reference uses from gym: the id registry and the `.env` + step-limit wrapper.
The dynamics follow this repo's FLOCK-SPEC v1 and run in HIP kernels (`mgp_flock_step`,
`mgp_flock_controller`): `VecFlock` is the B-episode device-resident simulator, the gym-style classes
are B = 1 facades over it that return numpy tuples.
"""
from dataclasses import dataclass, replace

import numpy as np
import torch

from .. import ops, _lib
from .._lib import MgpFlockParams


@dataclass(frozen=True)
class FlockParams:
    n_agents: int = 100
    comm_radius: float = 1.0
    v_max: float = 3.0
    v_bias: float = 3.0
    dt: float = 0.01
    max_rad_init: float = 1.0
    action_gain: float = 10.0
    max_accel: float = 1.0
    ctrl_gain: float = 0.1
    ctrl_clip: float = 10.0
    min_dist_thresh: float = 0.1
    min_degree: int = 2
    reward_scale: float = 1.0
    mean_pooling: bool = True
    max_episode_steps: int = 500
    n_leaders: int = 0
    two_flocks: bool = False
    init_mode: str = 'auto'
    grid_spacing: float = 0.6
    grid_jitter: float = 0.1
    centralized: bool = True     # controller() default: velocity consensus over ALL agents (the DAGGER teacher)
    link_drop: float = 0.0       # P(a radius link is down at a step) -- FlockingStochastic-v0 (FLOCK-SPEC item 8)
    link_seed: int = 0           # mixed into the fade hash (train.py passes the experiment seed)

    @property
    def link_drop_q32(self):
        """Drop threshold of the 32-bit fade hash: floor(link_drop * 2^32), saturated."""
        return max(0, min(0xFFFFFFFF, int(np.floor(float(self.link_drop) * 4294967296.0))))

    @property
    def symmetric_network(self):
        """Membership of the network is symmetric (j in N(i) <=> i in N(j)): true for every variant of FLOCK-SPEC v1 -- the
        radius test is, and link fading hashes the unordered pair.  The frame replay REQUIRES it (mgp_replay_gather walks bit
        ROW n as column n of A); a directed network (e.g. k nearest neighbours) must return False here, which keeps it off
        the frame-collecting path."""
        return True

    @property
    def comm_radius2(self):
        return self.comm_radius * self.comm_radius

    @property
    def r_max(self):
        return self.max_rad_init * np.sqrt(self.n_agents)

    def to_c(self):
        return MgpFlockParams(self.comm_radius2, self.dt, self.action_gain, self.max_accel, self.ctrl_gain,
                              self.ctrl_clip, self.reward_scale, 1 if self.mean_pooling else 0, self.n_leaders,
                              1 if self.centralized else 0, self.link_drop_q32, int(self.link_seed) & 0xFFFFFFFF, 0)


# ----------------------------------------------------------------------------------- reset sampling
def _sample_candidate(rng, p):
    """One draw of the reset distribution (FLOCK-SPEC v1 section 3).  RNG call order is part of the spec."""
    n = p.n_agents
    x = np.zeros((n, 4), dtype=np.float64)
    # two flocks: each half is drawn in a disc of HALF the area (same agent density as one flock of N, so the
    # acceptance test stays feasible), the discs sit side by side one communication radius apart
    area = p.r_max * (0.5 if p.two_flocks else 1.0)
    length = np.sqrt(rng.uniform(0, p.r_max, size=(n,)) * (area / p.r_max))
    angle = np.pi * rng.uniform(0, 2, size=(n,))
    x[:, 0] = length * np.cos(angle)
    x[:, 1] = length * np.sin(angle)
    bias = rng.uniform(low=-p.v_bias, high=p.v_bias, size=(2,))
    x[:, 2] = rng.uniform(low=-p.v_max, high=p.v_max, size=(n,)) + bias[0]
    x[:, 3] = rng.uniform(low=-p.v_max, high=p.v_max, size=(n,)) + bias[1]
    if p.two_flocks:
        half = n // 2
        shift = np.sqrt(area) + 0.5 * p.comm_radius
        x[:half, 0] -= shift
        x[half:, 0] += shift
        x[:half, 2] = x[:half, 2] - bias[0] + abs(bias[0])        # the two flocks head for each other
        x[half:, 2] = x[half:, 2] - bias[0] - abs(bias[0])
    return x


def lattice_sites(n):
    """First n integer lattice points ordered by (a^2+b^2, a, b): a disc-shaped patch of a square lattice."""
    m = int(np.ceil(np.sqrt(n / np.pi))) + 2
    a, b = np.meshgrid(np.arange(-m, m + 1), np.arange(-m, m + 1), indexing='ij')
    a, b = a.ravel(), b.ravel()
    order = np.lexsort((b, a, a * a + b * b))[:n]
    return np.stack([a[order], b[order]], axis=1).astype(np.float64)


def use_grid(p):
    return p.init_mode == 'grid' or (p.init_mode == 'auto' and p.n_agents > 100)


def _sample_candidate_grid(rng, p):
    """Grid-mode draw (FLOCK-SPEC v1 section 3b): jittered square lattice, pitch grid_spacing*R, jitter
    +-grid_jitter*R.  RNG call order: jitter_x(N) ; jitter_y(N) ; bias(2) ; vx(N) ; vy(N)."""
    n = p.n_agents
    x = np.zeros((n, 4), dtype=np.float64)
    sites = lattice_sites(n)
    s = p.grid_spacing * p.comm_radius
    j = p.grid_jitter * p.comm_radius
    x[:, 0] = sites[:, 0] * s + rng.uniform(-j, j, size=(n,))
    x[:, 1] = sites[:, 1] * s + rng.uniform(-j, j, size=(n,))
    bias = rng.uniform(low=-p.v_bias, high=p.v_bias, size=(2,))
    x[:, 2] = rng.uniform(low=-p.v_max, high=p.v_max, size=(n,)) + bias[0]
    x[:, 3] = rng.uniform(low=-p.v_max, high=p.v_max, size=(n,)) + bias[1]
    if p.two_flocks:                                              # lattice cut at x = 0, halves one radius apart, head-on
        left = sites[:, 0] < 0
        x[left, 0] -= 0.5 * p.comm_radius
        x[~left, 0] += 0.5 * p.comm_radius
        x[left, 2] = x[left, 2] - bias[0] + abs(bias[0])
        x[~left, 2] = x[~left, 2] - bias[0] - abs(bias[0])
    return x


def _candidate_ok(x, p):
    pos = x[:, 0:2]
    d = pos[:, None, :] - pos[None, :, :]
    r2 = d[:, :, 0] * d[:, :, 0] + d[:, :, 1] * d[:, :, 1]
    np.fill_diagonal(r2, np.inf)
    return (r2 < p.comm_radius2).sum(axis=1).min() >= p.min_degree and np.sqrt(r2.min()) >= p.min_dist_thresh


def sample_initial_state(rng, p, max_tries=100000):
    """Rejection-sample x0 (N,4): min degree >= p.min_degree and min distance >= p.min_dist_thresh.
    Host-side control logic, executed once per episode (the per-step arithmetic is on the device)."""
    for _ in range(max_tries):
        x = _sample_candidate_grid(rng, p) if use_grid(p) else _sample_candidate(rng, p)
        if _candidate_ok(x, p):
            return x
    raise RuntimeError("flock reset: no admissible initial configuration found")


def _candidates_from_uniforms(U, p):
    """(M, 4 n + 2) raw uniforms of the generator's stream -> (M, n, 4) candidates: _sample_candidate for every row of U, value
    for value (rng.uniform(low, high) is low + (high - low) * u on the same stream, element by element)."""
    n = p.n_agents
    M = U.shape[0]
    x = np.zeros((M, n, 4), dtype=np.float64)
    area = p.r_max * (0.5 if p.two_flocks else 1.0)
    length = np.sqrt((0 + (p.r_max - 0) * U[:, 0:n]) * (area / p.r_max))
    angle = np.pi * (0 + (2 - 0) * U[:, n:2 * n])
    x[:, :, 0] = length * np.cos(angle)
    x[:, :, 1] = length * np.sin(angle)
    span = p.v_bias - (-p.v_bias)
    bias = -p.v_bias + span * U[:, 2 * n:2 * n + 2]
    vspan = p.v_max - (-p.v_max)
    x[:, :, 2] = (-p.v_max + vspan * U[:, 2 * n + 2:3 * n + 2]) + bias[:, 0:1]
    x[:, :, 3] = (-p.v_max + vspan * U[:, 3 * n + 2:4 * n + 2]) + bias[:, 1:2]
    if p.two_flocks:
        half = n // 2
        shift = np.sqrt(area) + 0.5 * p.comm_radius
        x[:, :half, 0] -= shift
        x[:, half:, 0] += shift
        x[:, :half, 2] = x[:, :half, 2] - bias[:, 0:1] + np.abs(bias[:, 0:1])
        x[:, half:, 2] = x[:, half:, 2] - bias[:, 0:1] - np.abs(bias[:, 0:1])
    return x


def _candidate_positions(U, p):
    """The position half of _candidates_from_uniforms: (M, n, 2), contiguous."""
    n = p.n_agents
    area = p.r_max * (0.5 if p.two_flocks else 1.0)
    length = np.sqrt((0 + (p.r_max - 0) * U[:, 0:n]) * (area / p.r_max))
    angle = np.pi * (0 + (2 - 0) * U[:, n:2 * n])
    pos = np.empty((U.shape[0], n, 2), dtype=np.float64)
    pos[:, :, 0] = length * np.cos(angle)
    pos[:, :, 1] = length * np.sin(angle)
    if p.two_flocks:
        half = n // 2
        shift = np.sqrt(area) + 0.5 * p.comm_radius
        pos[:, :half, 0] -= shift
        pos[:, half:, 0] += shift
    return pos


def sample_initial_states(rng, p, B, device, max_tries=100000):
    """B consecutive sample_initial_state(rng, p) draws -- the same states from the same stream, the generator left where B
    sequential calls would leave it -- with the acceptance test of the candidates on the device (mgp_flock_reset_check).
    The rejection loop accepts about one disc draw in 140 at N = 100; its numpy test took 19 ms of host time per episode
    reset, most of the wall time of a training run.  Here a block of candidates is built from one block of the generator's
    raw uniforms (same values: see _candidates_from_uniforms), tested in one launch, and the generator is rewound and advanced
    to just behind the B-th accepted candidate.  Lattice resets (accepted at the first try) and generators without
    get_state / set_state take the sequential path."""
    if (use_grid(p) or B <= 0 or not all(hasattr(rng, a) for a in ('get_state', 'set_state', 'random_sample'))):
        return np.stack([sample_initial_state(rng, p, max_tries) for _ in range(B)]) if B > 0 else np.zeros((0, p.n_agents, 4))
    import ctypes
    n = p.n_agents
    per = 4 * n + 2
    L = _lib.lib()
    out = []
    tried = 0
    # candidates per block: bounded by a byte budget (M x (4 n + 2) fp64 uniforms <= 32 MB: at n = 1000 a block of 4096 would
    # be 130 MB on the host plus 65 MB on the device)
    m_cap = int(max(16, min(4096, (32 << 20) // (8 * per))))
    M = int(min(m_cap, max(256, 64 * B)))
    while True:
        state = rng.get_state()
        U = rng.random_sample((M, per))
        pos = torch.from_numpy(_candidate_positions(U, p)).to(device)        # (velocities: only for the accepted rows, below)
        deg = torch.empty((M,), device=device, dtype=torch.int32)
        r2m = torch.empty((M,), device=device, dtype=torch.float64)
        _lib.check(L.mgp_flock_reset_check(pos.data_ptr(), M, n, ctypes.c_double(p.comm_radius2), deg.data_ptr(), r2m.data_ptr(),
                                           ops._stream()), 'mgp_flock_reset_check')
        # (one pinned, non-blocking copy of both statistics and one wait, instead of two blocking .cpu() calls)
        host = torch.empty((2, M), dtype=torch.float64, pin_memory=torch.cuda.is_available())
        host[0].copy_(deg.to(torch.float64), non_blocking=True)
        host[1].copy_(r2m, non_blocking=True)
        torch.cuda.current_stream().synchronize()
        hn = host.numpy()
        ok = np.flatnonzero((hn[0] >= p.min_degree) & (np.sqrt(hn[1]) >= p.min_dist_thresh))
        take = ok[:B - len(out)]
        if take.size:
            out += list(_candidates_from_uniforms(U[take], p))
        if len(out) == B:
            used = int(take[-1]) + 1                              # candidates of this block the sequential loop would have drawn
            rng.set_state(state)
            rng.random_sample((used, per))
            return np.stack(out)
        tried += M
        if tried >= max_tries * B:
            raise RuntimeError("flock reset: no admissible initial configuration found")
        M = int(min(m_cap, 2 * M))


# ----------------------------------------------------------------------------------- device simulator
class VecFlock(object):
    """B independent flocking episodes, state and observations resident on one MI355X.

    Buffers (all preallocated, reused every step -> HIP-graph capturable):
      x (B,N,4) f64 | network (B,N,N) f32 | features (B,6,N) f32 (already (F,N)) | reward (B) f64
    """

    def __init__(self, B, params, device='cuda', want_f64_obs=False, with_expert=False):
        self.B, self.p = B, params
        self.with_expert = with_expert
        self.device = torch.device(device)
        if self.device.type != 'cuda':
            raise ops.MgpError("VecFlock needs a HIP device (no CPU simulation path)")
        N = params.n_agents
        self.N = N
        self._c = params.to_c()
        self.x = torch.zeros((B, N, 4), device=self.device, dtype=torch.float64)      # current state
        self._x_next = torch.zeros_like(self.x)                                        # ping-pong partner
        self._network, self._network_lazy = None, None
        self._network_own = torch.zeros((B, N, N), device=self.device, dtype=torch.float32)
        self._features_own = torch.zeros((B, 6, N), device=self.device, dtype=torch.float32)
        # `network` / `features` normally are the two buffers above; step(A_out=, feat_out=) / step_advance / the resident
        # rollout rebind them to (batch-strided) slots of a BatchedDelayState, refresh() binds them back
        self.network, self.features = self._network_own, self._features_own
        self.reward = torch.zeros((B,), device=self.device, dtype=torch.float64)
        self.expert = torch.zeros((B, N, 2), device=self.device, dtype=torch.float32)
        self.network64 = torch.zeros((B, N, N), device=self.device, dtype=torch.float64) if want_f64_obs else None
        self.features64 = torch.zeros((B, N, 6), device=self.device, dtype=torch.float64) if want_f64_obs else None
        self.expert64 = torch.zeros((B, N, 2), device=self.device, dtype=torch.float64) if want_f64_obs else None

    @property
    def network(self):
        """(B,N,N) fp32 network matrices of the current state.  After a resident rollout that left the dense operator
        unmaterialised this resolves through the delay state (which rebuilds its slices on first use)."""
        if self._network_lazy is not None:
            return self._network_lazy()
        return self._network

    @network.setter
    def network(self, value):
        self._network, self._network_lazy = value, None

    def set_state(self, x0):
        """x0 (B,N,4) array-like fp64: install states and refresh observations."""
        x0 = torch.as_tensor(np.asarray(x0, dtype=np.float64))
        assert x0.shape == (self.B, self.N, 4)
        self.x.copy_(x0)
        self.refresh()

    def reset(self, rng=None):
        rng = rng if rng is not None else np.random
        self.set_state(sample_initial_states(rng, self.p, self.B, self.device))

    def refresh(self):
        """Recompute observations (and the expert action, `params.centralized`) for the current x, no integration.
        The observations land in the simulator's OWN contiguous buffers: after a step that wrote them into the slots of a
        delay state (`step(A_out=...)`, `step_advance`, the resident rollout) `network` / `features` are strided views of
        that state's buffers, and a reset observation must not be written into -- or later pushed from -- those."""
        self.network, self.features = self._network_own, self._features_own
        ops.flock_step(self.x, None, self._c, A=self.network, A64=self.network64, feat=self.features,
                       feat64=self.features64, reward=self.reward, expert=self.expert if self.with_expert else None)

    def step(self, u, A_out=None, feat_out=None):
        """u (B,N,2) or the Actor output (B,1,2,N), fp32 on device.  Advances every episode one step in place;
        with `with_expert` the expert action of the new state (`params.centralized`) lands in `self.expert` for free.
        A_out / feat_out: optional (possibly batch-strided) destinations for the network matrix and the features,
        e.g. BatchedDelayState.next_slots(); self.network / self.features then alias them."""
        # without explicit destinations the observations go to the simulator's own buffers (never into slots of a delay
        # state that an earlier step_advance / resident rollout left `network` / `features` bound to: that state's
        # current slice 1 must stay A_t until the state itself advances)
        self.network = A_out if A_out is not None else self._network_own
        self.features = feat_out if feat_out is not None else self._features_own
        ops.flock_step(self.x, u, self._c, A=self.network, A64=self.network64, feat=self.features,
                       feat64=self.features64, reward=self.reward, expert=self.expert if self.with_expert else None,
                       x_out=self._x_next if u is not None else None)
        if u is not None:
            self.x, self._x_next = self._x_next, self.x          # ping-pong: several workgroups per episode

    def step_advance(self, u, state):
        """sim step + `state` (BatchedDelayState) transition: ONE fused kernel when the shape allows it
        (mgp_flock_step_advance), else the two-call in-place protocol.  Leaves `state` advanced."""
        A_dst, X_dst = state.next_slots()
        fused = (self.network64 is None and self.features64 is None and state.F == 6 and u is not None
                 and u.is_contiguous() and state.K > 1)
        if fused:
            G_prev, G_next, Xd_prev, Xd_next = state.buffers()
            fused = ops.flock_step_advance(self.x, self._x_next, u, self._c, G_prev, G_next, Xd_prev, Xd_next,
                                           state.has_prev, reward=self.reward,
                                           expert=self.expert if self.with_expert else None)
        if fused:
            self.x, self._x_next = self._x_next, self.x
            self.network, self.features = A_dst, X_dst
            state.flip()
        else:
            self.step(u, A_out=A_dst, feat_out=X_dst)
            state.advance()

    def controller(self, centralized=None):
        """Expert action for the current state -> (B,N,2) fp32 (buffer reused).  None = `params.centralized`."""
        if centralized is None:
            centralized = self.p.centralized
        if self.with_expert and bool(centralized) == bool(self.p.centralized) and self.expert64 is None:
            return self.expert                      # already produced by the last step()/refresh()
        ops.flock_controller(self.x, self._c, centralized=bool(centralized), u=self.expert, u64=self.expert64)
        return self.expert


# ----------------------------------------------------------------------------------- gym-style facade
class DeviceObservation(object):
    """One half of the env's observation tuple that stays on the device until somebody asks for numpy.

    Behaves like the fp64 numpy array the reference expects (`shape`, `dtype`, `np.asarray(obs)`, indexing,
    `transpose`, arithmetic through `__array__`), but `MultiAgentStateWithDelay` recognises it and takes `.device32`
    -- the fp32 tensor already on the GPU in the layout it needs -- so the per-step D2H of the 80 KB fp64 network
    matrix and the H2D of its fp32 copy disappear from the reference-style B = 1 loop."""

    def __init__(self, dev64, device32, zero_diagonal=False, shape=None):
        self._dev64 = dev64                  # (N,6) or (N,N) float64 device tensor (snapshot); None in the fast loop mode
        self.device32 = device32             # (6,N) features / (N,N) network, float32, device
        self.zero_diagonal = zero_diagonal   # the simulator never sets self-loops (state_with_delay.py:26)
        self._shape = tuple(shape) if shape is not None else tuple(dev64.shape)
        self._np = None

    @property
    def shape(self):
        return self._shape

    @property
    def dtype(self):
        return np.dtype(np.float64)

    @property
    def ndim(self):
        return len(self._shape)

    def numpy(self):
        if self._np is None:
            if self._dev64 is None:
                raise ops.MgpError("this observation was produced in the environment's fast loop mode (fast_loop = True): only "
                                   "its fp32 device side exists; set env.env.fast_loop = False for numpy observations")
            self._np = self._dev64.cpu().numpy()
        return self._np

    def __array__(self, dtype=None, copy=None):
        a = self.numpy()
        return a if dtype is None else a.astype(dtype, copy=False)

    def __len__(self):
        return self.shape[0]

    def __getitem__(self, idx):
        return self.numpy()[idx]

    def __getattr__(self, name):             # transpose / reshape / sum / ... : defer to the numpy view
        if name.startswith('_'):
            raise AttributeError(name)
        if self._np is None and self._dev64 is None:
            # fast loop mode: there is no numpy view to defer to -- an AttributeError, so that hasattr() / getattr(obj, name,
            # default) behave (np.asarray(obs) still raises the MgpError that explains the mode)
            raise AttributeError("%s (observation of the environment's fast loop mode: only its fp32 device side exists)" % name)
        return getattr(self.numpy(), name)


class FlockingRelativeEnv(object):
    """Single-episode environment with the raw-env interface the reference reaches through `env.env`."""

    variant = {}

    def __init__(self, device=None):
        self.params = replace(FlockParams(), **self.variant)
        self.device = device
        self._sim = None
        self._rng = np.random          # the reference seeds numpy's global RNG (train.py:27)
        self.n_features = 6
        self.nu = 2
        self.lazy_obs = True           # observations stay on the device until numpy is requested
        # fast loop mode (set by this package's own one-environment loops, learner/imitation.py / rollouts.py): nothing
        # crosses PCIe per step -- observations carry only their fp32 device side, step() takes the action as a device
        # tensor and returns the reward as a 0-d device tensor, controller() returns a device tensor.  The values are the
        # ones the default mode returns (the device consumes actions as fp32 either way); off by default because code
        # written against gym_flock expects numpy.
        self._fast_loop = False

    @property
    def fast_loop(self):
        return self._fast_loop

    @fast_loop.setter
    def fast_loop(self, on):
        if bool(on) != self._fast_loop:
            self._fast_loop = bool(on)
            old = self._sim
            self._sim = None
            if old is not None:                      # mid-episode toggle: the new simulator continues from the same state
                self._ensure().set_state(old.x.cpu().numpy())

    # -- configuration -------------------------------------------------------------------------
    def params_from_cfg(self, args):
        kw = dict(n_agents=args.getint('n_agents'), comm_radius=args.getfloat('comm_radius'),
                  v_max=args.getfloat('v_max'), v_bias=args.getfloat('v_max'))
        if args.get('dt') is not None:                      # absent in the reference's *_stoch.cfg files
            kw['dt'] = args.getfloat('dt')
        if args.get('link_drop') is not None:               # this package's key; the variant's default otherwise
            kw['link_drop'] = args.getfloat('link_drop')
        self.params = replace(self.params, **kw)
        self._sim = None

    def seed(self, seed=None):
        self._rng = np.random.RandomState(seed) if seed is not None else np.random
        if seed is not None and self.params.link_drop > 0.0:
            self.params = replace(self.params, link_seed=int(seed) & 0xFFFFFFFF)
            self._sim = None
        return [seed]

    @property
    def n_agents(self):
        return self.params.n_agents

    def _ensure(self):
        if self._sim is None:
            dev = self.device or ('cuda:0' if torch.cuda.is_available() else None)
            if dev is None:
                raise ops.MgpError("the flocking simulator needs a HIP device (no CPU simulation path)")
            self._sim = VecFlock(1, self.params, dev, want_f64_obs=not self._fast_loop, with_expert=self._fast_loop)
        return self._sim

    def _obs(self):
        s = self._sim
        if self._fast_loop:
            n = self.params.n_agents
            return (DeviceObservation(None, s.features[0].clone(), shape=(n, 6)),
                    DeviceObservation(None, s.network[0].clone(), zero_diagonal=True, shape=(n, n)))
        if not self.lazy_obs:
            return s.features64[0].cpu().numpy(), s.network64[0].cpu().numpy()
        # snapshots: the simulator's buffers are overwritten by the next step
        return (DeviceObservation(s.features64[0].clone(), s.features[0].clone()),
                DeviceObservation(s.network64[0].clone(), s.network[0].clone(), zero_diagonal=True))

    # -- gym API -------------------------------------------------------------------------------
    def reset(self):
        s = self._ensure()
        s.reset(self._rng)
        return self._obs()

    def step(self, u):
        s = self._ensure()
        if torch.is_tensor(u) and u.is_cuda:                    # action already on the device (this package's loops)
            assert tuple(u.shape) == (self.params.n_agents, self.nu)
            ut = u.to(torch.float32).reshape(1, -1, 2).contiguous()
        else:
            u = np.asarray(u.cpu() if torch.is_tensor(u) else u)
            assert u.shape == (self.params.n_agents, self.nu)
            ut = torch.from_numpy(np.ascontiguousarray(u, dtype=np.float32)).reshape(1, -1, 2).to(s.device)
        s.step(ut)
        if self._fast_loop:
            return self._obs(), s.reward[0].clone(), False, {}
        return self._obs(), float(s.reward[0].item()), False, {}

    def controller(self, centralized=None):
        """None -> the env's own `centralized` attribute (True: DAGGER's teacher is the global controller)."""
        s = self._ensure()
        u = s.controller(self.params.centralized if centralized is None else bool(centralized))
        if self._fast_loop:
            return u[0].clone()                                 # (N,2) fp32 on the device
        return s.expert64[0].cpu().numpy()

    def render(self, mode='human'):
        return None

    def close(self):
        self._sim = None


class FlockingLeaderEnv(FlockingRelativeEnv):
    """FLOCK-SPEC v1 variant: the first `n_leaders` agents keep their initial velocity (ignore u)."""
    variant = dict(n_leaders=2)


class FlockingTwoFlocksEnv(FlockingRelativeEnv):
    """FLOCK-SPEC v1 variant: reset draws two groups displaced along x with opposite velocity bias."""
    variant = dict(two_flocks=True)


class FlockingStochasticEnv(FlockingRelativeEnv):
    """FLOCK-SPEC v1 variant: every radius link is down with probability `link_drop` at each step (position-keyed
    fading, item 8 of the spec) -- network rows, features and the decentralised controller see the faded graph."""
    variant = dict(link_drop=0.1)


class TimeLimit(object):
    """The part of gym.wrappers.TimeLimit the reference relies on: `.env`, and `done` after
    `max_episode_steps` steps (reference gnn_dagger.py:154,163: `while not done`)."""

    def __init__(self, env, max_episode_steps):
        self.env = env
        self._max_episode_steps = max_episode_steps
        self._elapsed = 0

    def seed(self, seed=None):
        return self.env.seed(seed)

    def reset(self):
        self._elapsed = 0
        return self.env.reset()

    def step(self, action):
        obs, reward, done, info = self.env.step(action)
        self._elapsed += 1
        if self._elapsed >= self._max_episode_steps:
            done = True
            info = dict(info)
            info['TimeLimit.truncated'] = True
        return obs, reward, done, info

    def render(self, mode='human'):
        return self.env.render(mode)

    def close(self):
        return self.env.close()


_REGISTRY = {
    'FlockingRelative-v0': FlockingRelativeEnv,
    'FlockingLeader-v0': FlockingLeaderEnv,
    'FlockingTwoFlocks-v0': FlockingTwoFlocksEnv,
    'FlockingStochastic-v0': FlockingStochasticEnv,
}


def registered_ids():
    return sorted(_REGISTRY)


def make(name, device=None, max_episode_steps=None):
    """`gym.make` stand-in for the ids the reference's cfg files name."""
    if name not in _REGISTRY:
        raise KeyError("unknown environment id %r (known: %s)" % (name, ', '.join(registered_ids())))
    env = _REGISTRY[name](device=device)
    return TimeLimit(env, max_episode_steps or env.params.max_episode_steps)
