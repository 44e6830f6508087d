"""GPU: the factored-state path for flocks beyond the LDS-resident kernel (N > 256): mgp_flock_step_sparse,
mgp_sparse_policy_step, mgp_sparse_to_dense against the oracle and against the dense two-launch path."""
import numpy as np
import pytest
import torch

from oracle import actor as oa, flock as ofl, state as os_
from test_gpu_rollout import _make, _snapshot, _weights_np, relerr, elem_err
from conftest import reference_noise, check_parity

pytestmark = pytest.mark.gpu

CASES = [
    # N, K, hidden, variant
    (300, 3, (32, 32), {}),
    (300, 4, (32,), {'mean_pooling': False, 'n_leaders': 2}),
    (257, 2, (16, 16), {'comm_radius': 1.5}),
    (513, 3, (32,), {'link_drop': 0.3, 'link_seed': 4}),
    (300, 1, (32, 32), {}),
    (320, 5, (16,), {}),
    (1000, 3, (32, 32), {}),
]


def _bits_to_dense(bits, wrow, N):
    """(N, NW) int64 rows + (N,) weights -> dense (N, N) fp32 network matrix."""
    b = bits.astype(np.uint64)
    cols = np.arange(N)
    pat = ((b[:, cols >> 6] >> (cols & 63).astype(np.uint64)) & np.uint64(1)).astype(np.float32)
    return pat * wrow[:, None].astype(np.float32)


@pytest.mark.parametrize('N,K,hidden,variant', CASES)
def test_sparse_rollout_matches_oracle_step_by_step(N, K, hidden, variant):
    """From a reset: every step's action against the oracle forward on the oracle's own dense state, the simulator outputs
    (bit rows, weights, features, state) against the oracle transition, and the dense state rebuilt by to_dense."""
    from multiagent_gnn_policies_amd.learner.sparse_rollout import SparseFlockState, sparse_policy_rollout, sparse_supported
    B = 2
    rs, op, actor, sim, st = _make(N, K, hidden, B, seed=N + K, **variant)
    assert sparse_supported(actor, K, N)
    Ws, bs = _weights_np(actor)
    x0, G, X = _snapshot(sim, st)                                        # oracle-side dense state, advanced by the oracle below
    G = G.astype(np.float64); X = X.astype(np.float64)
    sp = SparseFlockState(sim, K)
    sp.observe_reset(sim)
    xs = x0.copy()
    action = torch.zeros((B, 1, 2, N), device='cuda')
    rewards = torch.zeros((B, 1), device='cuda', dtype=torch.float64)
    for step in range(K + 2):
        # the factored state of the current step vs the oracle's network / features
        for b in range(B):
            h = ofl.helpers(xs[b], op)
            got = _bits_to_dense(sp.bits[b, sp.hs].cpu().numpy(), sp.wrow[b, sp.hs].cpu().numpy(), N)
            assert np.array_equal(got, h['network'].astype(np.float32)), "network bits / weights must be exact"
            assert relerr(sp.feat[b, sp.cur, :, :6].cpu().numpy(), h['values'].astype(np.float32)) <= 1e-6
        noise, ref = reference_noise(X.astype(np.float32), G.astype(np.float32), Ws, bs, K, per_episode=True)
        sparse_policy_rollout(actor, sim, sp, 1, rewards=rewards, action=action)
        u = action.cpu().numpy()
        check_parity(u, ref, noise, 'factored path, step %d' % step)                        # elementwise, conftest.NOISE_FACTOR
        for b in range(B):
            x2, vals, net, r = ofl.step(xs[b], u[b, 0].T.astype(np.float32), op)
            assert np.array_equal(sim.x[b].cpu().numpy(), x2), "integration must be bit-exact fp64 given the action"
            assert abs(rewards[b, 0].item() - r) <= 1e-12 * max(1.0, abs(r))
            xs[b] = x2
            Gn, Xn = os_.gso_update(net[None], G[b:b + 1].astype(np.float32), vals.T[None].astype(np.float32),
                                    X[b:b + 1].astype(np.float32), K, dtype=np.float64)
            G[b], X[b] = Gn[0], Xn[0]
    sp.to_dense(sim, st)
    x1, G1, X1 = _snapshot(sim, st)
    assert relerr(G1, G) <= 1e-6 and relerr(X1, X) <= 1e-6
    assert np.array_equal(G1[:, 0], np.broadcast_to(np.eye(N, dtype=np.float32), (B, N, N)))


@pytest.mark.parametrize('N,K', [(300, 3), (400, 4), (260, 2)])
def test_policy_rollout_takes_the_sparse_path_and_chunks_exactly(N, K):
    """policy_rollout from a reset observation: the factored path runs (True), agrees with the dense two-launch path over
    a few steps, and -- the factored state being carried between calls -- chunked calls are bit-identical."""
    from multiagent_gnn_policies_amd.learner.rollouts import policy_rollout
    B, T = 2, 6
    outs = []
    for mode in ('sparse', 'sparse_chunked', 'dense'):
        rs, op, actor, sim, st = _make(N, K, (32, 32), B, seed=5)
        rewards = torch.zeros((B, T), device='cuda', dtype=torch.float64)
        action = torch.zeros((B, 1, 2, N), device='cuda')
        if mode == 'sparse_chunked':
            r1 = torch.zeros((B, 2), device='cuda', dtype=torch.float64); r2 = torch.zeros((B, 4), device='cuda', dtype=torch.float64)
            assert policy_rollout(actor, sim, st, 2, rewards=r1, action=action)
            assert policy_rollout(actor, sim, st, 4, rewards=r2, action=action)
            rewards[:, :2] = r1; rewards[:, 2:] = r2
        else:
            ran = policy_rollout(actor, sim, st, T, rewards=rewards, action=action, resident=(mode == 'sparse'))
            assert ran == (mode == 'sparse')
        outs.append(_snapshot(sim, st) + (action.cpu().numpy().copy(), rewards.cpu().numpy().copy()))
    for a, b in zip(outs[0], outs[1]):
        assert np.array_equal(a, b)                                       # carried factored state: exact chunking
    for name, a, b, tol in zip(('x', 'G', 'X', 'u', 'r'), outs[0], outs[2], (1e-5, 1e-3, 1e-3, 1e-3, 1e-5)):
        assert relerr(a, b) <= tol, (name, relerr(a, b))
    # mid-episode without a carried factored state: the dense path takes over
    rs, op, actor, sim, st = _make(N, K, (32, 32), B, seed=5)
    assert policy_rollout(actor, sim, st, 2, resident=False) is False
    assert policy_rollout(actor, sim, st, 2) is False


@pytest.mark.parametrize('N,spread,variant', [(1000, 1.0, {}), (300, 0.02, {}), (2048, 1.0, {'mean_pooling': False}),
                                              (777, 4.0, {'link_drop': 0.2, 'link_seed': 3}), (260, 1.0, {'n_leaders': 3})])
def test_cell_list_simulator_equals_the_all_pairs_kernel(N, spread, variant):
    """mgp_flock_step_cells vs mgp_flock_step_sparse on the same states: identical bit rows, weights and integration;
    feature sums differ only in the order of their fp64 terms.  spread 0.02 collapses the flock into one cell, 4.0
    stretches it along x beyond the 64-cell limit."""
    from multiagent_gnn_policies_amd.envs import FlockParams, VecFlock
    from multiagent_gnn_policies_amd.learner.sparse_rollout import SparseFlockState
    B = 2
    rs = np.random.RandomState(N)
    op = ofl.FlockParams(n_agents=N, init_mode='grid', **variant)
    p = FlockParams(**{f: getattr(op, f) for f in FlockParams.__dataclass_fields__})
    xs = np.stack([ofl.sample_candidate_grid(rs, op) for _ in range(B)])
    if spread > 1.0:
        xs[:, :, 0] *= spread                 # stretched along x only: beyond 64 cells of width R, neighbours along y remain
    else:
        xs[:, :, :2] *= spread
    res = []
    for cells in (True, False):
        sim = VecFlock(B, p, 'cuda', with_expert=True)
        sim.x.copy_(torch.from_numpy(xs))
        sp = SparseFlockState(sim, 3)
        sp.use_cells = cells
        sp.observe_reset(sim)
        u = torch.from_numpy(np.random.RandomState(1).uniform(-1.2, 1.2, size=(B, 1, 2, N)).astype(np.float32)).cuda()
        sp.step(sim, u)
        sp.step(sim, u)
        res.append((sim.x.cpu().numpy().copy(), sp.bits.cpu().numpy().copy(), sp.wrow.cpu().numpy().copy(),
                    sp.feat.cpu().numpy().copy(), sim.reward.cpu().numpy().copy(), sim.expert.cpu().numpy().copy()))
    a, b = res
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2])
    assert relerr(a[3], b[3]) <= 1e-6 and relerr(a[5], b[5]) <= 1e-6
    assert np.max(np.abs(a[4] - b[4]) / np.maximum(1.0, np.abs(b[4]))) <= 1e-12
    assert a[1].any()


@pytest.mark.parametrize('N,K,hidden', [(300, 3, (32, 32)), (1000, 3, (32, 32)), (400, 4, (32,)), (260, 2, (16, 16)), (320, 5, (16,)), (2600, 3, (32, 32))])
def test_staged_and_direct_gather_forms_agree(N, K, hidden):
    """mgp_sparse_policy_step has two forms of its gather / policy launches -- source rows staged in the LDS (the default
    where they fit) and gathered straight from global memory (the fallback for very large flocks).  On the bit rows both sum
    a column in the same order: bit-identical actions over a rollout from a reset.  The default additionally reads the
    networks as compact neighbour lists (a different, fixed order of the same terms): equal to fp32 rounding."""
    from multiagent_gnn_policies_amd import _lib
    from multiagent_gnn_policies_amd.learner.sparse_rollout import SparseFlockState, sparse_policy_rollout
    outs = {}
    for mode in (0, 1, 2):                                      # default (lists) | direct | staged on bit rows
        old = _lib.lib().mgp_sparse_force_direct(mode)
        try:
            rs, op, actor, sim, st = _make(N, K, hidden, 2, seed=N + K)
            sp = SparseFlockState(sim, K)
            sp.observe_reset(sim)
            action = torch.zeros((2, 1, 2, N), device='cuda')
            acts = []
            for step in range(K + 2):
                sparse_policy_rollout(actor, sim, sp, 1, action=action)
                acts.append(action.clone())
            outs[mode] = (torch.stack(acts), sim.x.clone())
        finally:
            _lib.lib().mgp_sparse_force_direct(old)
    assert torch.equal(outs[1][0], outs[2][0]) and torch.equal(outs[1][1], outs[2][1])
    a0, a1 = outs[0][0], outs[1][0]
    assert torch.isfinite(a0).all() and float(a0.abs().max()) > 0
    scale = max(1.0, float(a1.abs().max()))
    assert torch.equal(a0[0], a1[0])                             # the reset observation has no history: nothing to gather
    if K >= 2:
        assert float((a0[1] - a1[1]).abs().max()) <= 1e-6 * scale   # first gathered step: fp32 rounding of a re-ordered sum
    # later steps: the closed loop amplifies that rounding (measured 2e-6 relative after four steps at outputs of 39)
    assert float((a0 - a1).abs().max()) <= 1e-4 * scale
    assert float((outs[0][1] - outs[1][1]).abs().max()) <= 1e-4


def test_neighbour_lists_mirror_the_bit_rows():
    """mgp_flock_step_cells_nbr: every row's compact list holds exactly the set bits of its bit row (count at position 15,
    entry e at position (e & 3) * 4 + (e >> 2)), or the overflow mark when the row has more than 15 neighbours -- on a
    lattice (degrees 6-10) and on a contracted flock (degrees beyond the list)."""
    from multiagent_gnn_policies_amd.learner.sparse_rollout import SparseFlockState
    N, K, B = 700, 3, 2
    rs, op, actor, sim, st = _make(N, K, (32,), B, seed=3)
    for scale in (1.0, 0.45):
        x = sim.x.clone()
        x[:, :, :2] *= scale
        sim.x.copy_(x)
        sp = SparseFlockState(sim, K)
        sp.observe_reset(sim)
        assert sp._nbr_ok
        nb = sp.nbr[:, sp.hs].cpu().numpy().astype(np.uint16)
        bits = sp.bits[:, sp.hs].cpu().numpy().astype(np.uint64)
        cols = np.arange(N)
        n_list = n_over = 0
        for b in range(B):
            member = ((bits[b][:, cols >> 6] >> (cols & 63).astype(np.uint64)) & np.uint64(1)).astype(bool)
            for i in range(N):
                want = set(np.nonzero(member[i])[0].tolist())
                cnt = int(nb[b, i, 15])
                if cnt == 0xFFFF:
                    assert len(want) > 15
                    n_over += 1
                else:
                    got = [int(nb[b, i, (e & 3) * 4 + (e >> 2)]) for e in range(cnt)]
                    assert cnt == len(want) and set(got) == want and len(set(got)) == cnt
                    n_list += 1
        assert (n_over == 0 and n_list == B * N) if scale == 1.0 else n_over > 0


@pytest.mark.parametrize('N,hidden,B,scale,variant', [
    (300, (32, 32), 3, 1.0, {}),
    (1000, (32, 32), 2, 1.0, {}),
    (700, (32,), 2, 1.0, {'mean_pooling': False, 'n_leaders': 2}),
    (1000, (32, 32), 2, 0.45, {}),                 # contracted flock: degrees beyond the 15-entry lists (bit-row fallback)
    (520, (16, 32, 8), 2, 1.0, {'comm_radius': 1.4}),
    (513, (32,), 2, 1.0, {'link_drop': 0.3, 'link_seed': 4}),   # FlockingStochastic's link fading
    (513, (32,), 2, 0.4, {'link_drop': 0.2, 'link_seed': 9}),   # ... on the bit-row fallback
    (300, (32, 32), 131, 1.0, {}),                 # more episodes than one launch of resident workgroups holds (256 CUs / 2 tiles)
])
def test_persistent_factored_rollout_is_bit_identical(N, hidden, B, scale, variant, monkeypatch):
    """mgp_sparse_rollout at K = 3, N <= 1024 runs its T steps as ONE launch of persistent workgroups (csrc/sparse_persist.hip:
    the episode's rows stay in LDS, siblings exchange through the state buffers behind arrival counters).  Same arithmetic in
    the same order: actions, rewards, the fp64 state and every ring -- features, row weights, bit rows, list rows -- equal the
    K-launch form's bit for bit, in one call and chunked (odd first chunk: the second call starts from the other x buffer)."""
    import ctypes
    from multiagent_gnn_policies_amd import _lib
    from multiagent_gnn_policies_amd.learner.sparse_rollout import SparseFlockState, sparse_policy_rollout
    K, T = 3, 7
    outs = []
    for mode in ('launches', 'persistent', 'persistent_chunked'):
        monkeypatch.setenv('MGP_SP_PERSIST', '0' if mode == 'launches' else '1')
        rs, op, actor, sim, st = _make(N, K, hidden, B, seed=N + 1, **variant)
        dims = tuple(actor.layers)
        cd = (ctypes.c_int * len(dims))(*dims)
        assert _lib.lib().mgp_sparse_rollout_persistent(cd, len(dims) - 1, K, N, ctypes.byref(sim._c)) == (0 if mode == 'launches' else 1)
        if scale != 1.0:
            x = sim.x.clone(); x[:, :, :2] *= scale; sim.x.copy_(x)
        sp = SparseFlockState(sim, K)
        sp.observe_reset(sim)
        action = torch.zeros((B, 1, 2, N), device='cuda')
        rewards = torch.zeros((B, T), device='cuda', dtype=torch.float64)
        if mode == 'persistent_chunked':
            r1 = torch.zeros((B, 3), device='cuda', dtype=torch.float64); r2 = torch.zeros((B, 4), device='cuda', dtype=torch.float64)
            sparse_policy_rollout(actor, sim, sp, 3, rewards=r1, action=action)
            sparse_policy_rollout(actor, sim, sp, 4, rewards=r2, action=action)
            rewards[:, :3] = r1; rewards[:, 3:] = r2
        else:
            sparse_policy_rollout(actor, sim, sp, T, rewards=rewards, action=action)
        sp.check_status()
        assert torch.isfinite(action).all() and torch.isfinite(rewards).all() and float(action.abs().max()) > 0
        outs.append(dict(x=sim.x.clone(), action=action.clone(), rewards=rewards.clone(), bits=sp.bits.clone(), wrow=sp.wrow.clone(),
                         feat=sp.feat.clone(), nbr=sp.nbr.clone(), expert=sim.expert.clone() if sim.with_expert else None,
                         slots=(sp.cur, sp.hs)))
    if scale != 1.0:
        assert (outs[0]['nbr'][:, :, :, 15].to(torch.int32) & 0xFFFF).eq(0xFFFF).any(), "the contracted flock must overflow some lists"
    for other in outs[1:]:
        for k, v in outs[0].items():
            if k == 'slots':
                assert v == other[k]
            elif v is not None:
                assert torch.equal(v, other[k]), "%s differs between the K-launch and the persistent form" % k


def test_persistent_rollout_gives_up_loudly_when_a_sibling_never_arrives(monkeypatch):
    """An episode's persistent workgroups spin on each other's arrival counters; CUs held by another process could keep a
    sibling from ever starting.  Every poll is bounded: with the test hook muting one workgroup of episode 1 (from step 1 on)
    its siblings give up after the timeout, the episode's action / state / rewards are NaN, the status call reports it --
    and the other episodes of the same launch finish with exactly the results of an undisturbed run."""
    from multiagent_gnn_policies_amd import _lib
    from multiagent_gnn_policies_amd.learner.sparse_rollout import SparseFlockState, sparse_policy_rollout
    N, K, B, T = 300, 3, 3, 5
    monkeypatch.setenv('MGP_SP_PERSIST', '1')
    monkeypatch.setenv('MGP_SP_PERSIST_TIMEOUT_MS', '200')
    outs = []
    for fault in (None, '1'):
        if fault is None:
            monkeypatch.delenv('MGP_SP_PERSIST_FAULT', raising=False)
        else:
            monkeypatch.setenv('MGP_SP_PERSIST_FAULT', fault)
        rs, op, actor, sim, st = _make(N, K, (32, 32), B, seed=9)
        sp = SparseFlockState(sim, K)
        sp.observe_reset(sim)
        action = torch.zeros((B, 1, 2, N), device='cuda')
        rewards = torch.zeros((B, T), device='cuda', dtype=torch.float64)
        sparse_policy_rollout(actor, sim, sp, T, rewards=rewards, action=action)
        if fault is None:
            sp.check_status()
        else:
            with pytest.raises(_lib.MgpError):
                sp.check_status()
        outs.append((sim.x.clone(), action.clone(), rewards.clone()))
    (x0, a0, r0), (x1, a1, r1) = outs
    assert torch.isnan(a1[1]).all() and torch.isnan(x1[1]).all() and torch.isnan(r1[1]).all()
    for b in (0, 2):
        assert torch.equal(x0[b], x1[b]) and torch.equal(a0[b], a1[b]) and torch.equal(r0[b], r1[b])
    # the state object is usable again after a reset observation
    monkeypatch.delenv('MGP_SP_PERSIST_FAULT', raising=False)
    rs, op, actor, sim, st = _make(N, K, (32, 32), B, seed=9)
    sp = SparseFlockState(sim, K)
    sp.observe_reset(sim)
    sparse_policy_rollout(actor, sim, sp, T)
    sp.check_status()
    assert torch.equal(sim.x, x0)


def test_persistent_rollout_under_uneven_load(monkeypatch):
    """The sibling exchanges of the persistent form (arrival counters, L1-bypassing loads; plain stores where an episode's
    workgroups share an XCD: B a multiple of 8) with the chip busy on something else: a second stream streams copies through
    HBM while the rollout runs, so workgroups start late, siblings wait for each other and the caches are anything but cold.
    Every repetition must reproduce the K-launch form's result of the idle chip bit for bit."""
    from multiagent_gnn_policies_amd.learner.sparse_rollout import SparseFlockState, sparse_policy_rollout
    N, K, B, T = 1000, 3, 8, 40

    def run(persist, load):
        monkeypatch.setenv('MGP_SP_PERSIST', persist)
        rs, op, actor, sim, st = _make(N, K, (32, 32), B, seed=21)
        sp = SparseFlockState(sim, K)
        sp.observe_reset(sim)
        action = torch.zeros((B, 1, 2, N), device='cuda')
        rewards = torch.zeros((B, T), device='cuda', dtype=torch.float64)
        torch.cuda.synchronize()
        if load:
            side = torch.cuda.Stream()
            a = torch.empty((64 << 20,), device='cuda', dtype=torch.float32); b_ = torch.ones_like(a)
            with torch.cuda.stream(side):
                for _ in range(60):
                    a.copy_(b_); b_.add_(a, alpha=0.5)
        sparse_policy_rollout(actor, sim, sp, T, rewards=rewards, action=action)
        sp.check_status()
        torch.cuda.synchronize()
        return sim.x.clone(), action.clone(), rewards.clone(), sp.feat.clone(), sp.wrow.clone(), sp.bits.clone(), sp.nbr.clone()

    ref = run('0', False)
    for rep in range(3):
        got = run('1', True)
        for a, b_ in zip(ref, got):
            assert torch.equal(a, b_), "repetition %d differs from the K-launch form" % rep


def test_persistent_rollout_is_graph_capturable(monkeypatch):
    """mgp_sparse_rollout's persistent form inside a HIP graph (the counters' memset is a node of its own): captured once,
    replayed twice = two eager calls, bit for bit (T a multiple of 6: the ring slots and the x ping-pong are back in place,
    so the captured pointers describe every replay)."""
    from multiagent_gnn_policies_amd import ops
    from multiagent_gnn_policies_amd.learner.sparse_rollout import SparseFlockState, sparse_policy_rollout
    monkeypatch.setenv('MGP_SP_PERSIST', '1')
    N, K, B, T = 1000, 3, 8, 6
    outs = []
    for mode in ('eager', 'graph'):
        rs, op, actor, sim, st = _make(N, K, (32, 32), B, seed=3)
        sp = SparseFlockState(sim, K)
        sp.observe_reset(sim)
        action = torch.zeros((B, 1, 2, N), device='cuda')
        if mode == 'eager':
            sparse_policy_rollout(actor, sim, sp, T, action=action)
            sparse_policy_rollout(actor, sim, sp, T, action=action)
        else:
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            slots = (sp.cur, sp.hs)
            with ops.graph_capture(g):
                sparse_policy_rollout(actor, sim, sp, T, action=action)
            assert (sp.cur, sp.hs) == slots
            g.replay()
            g.replay()
        sp.check_status()
        outs.append((sim.x.clone(), action.clone(), sp.feat.clone(), sp.wrow.clone(), sp.bits.clone()))
    for a, b_ in zip(*outs):
        assert torch.equal(a, b_)


@pytest.mark.parametrize('N,B,tiles', [(1000, 2, 4), (1000, 3, 9), (300, 3, 2), (700, 2, 44)])
def test_persistent_rollout_tile_counts(N, B, tiles, monkeypatch):
    """How many workgroups share an episode is a launch decision (ceil(N / 256) at least, more when the call has fewer episodes
    than the device CUs; rows dealt evenly in whole 16-column tiles): every count gives the K-launch form's bits.  The
    small-batch cases of the other tests run on many small tiles; here the count is forced, 4 x 250 rows at N = 1000 (the
    layout of BASELINE configs[2]: 64 episodes on 256 CUs) included."""
    from multiagent_gnn_policies_amd.learner.sparse_rollout import SparseFlockState, sparse_policy_rollout
    K, T = 3, 5
    outs = []
    for persist in ('0', '1'):
        monkeypatch.setenv('MGP_SP_PERSIST', persist)
        monkeypatch.setenv('MGP_SP_PERSIST_TILES', str(tiles))
        rs, op, actor, sim, st = _make(N, K, (32, 32), B, seed=N + tiles)
        sp = SparseFlockState(sim, K)
        sp.observe_reset(sim)
        action = torch.zeros((B, 1, 2, N), device='cuda')
        rewards = torch.zeros((B, T), device='cuda', dtype=torch.float64)
        sparse_policy_rollout(actor, sim, sp, T, rewards=rewards, action=action)
        sp.check_status()
        outs.append((sim.x.clone(), action.clone(), rewards.clone(), sp.feat.clone(), sp.wrow.clone(), sp.bits.clone(), sp.nbr.clone()))
    for a, b_ in zip(*outs):
        assert torch.equal(a, b_)
