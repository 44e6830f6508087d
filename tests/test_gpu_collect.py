"""GPU: DAGGER data collection on the episode-resident kernel (mgp_rollout_collect) and the compact frame replay
(mgp_replay_gather) against the oracle: reference gnn_dagger.py:154-178 per lane -- expert label for the state before the
step, beta coin, expert- or policy-driven step, state transition -- with the coin spec of oracle/dagger_vec.py, and
state_with_delay.py:44-53 for the K-tap states rebuilt from frames."""
import numpy as np
import pytest
import torch

from oracle import actor as oa, dagger_vec as odv, flock as ofl
from test_gpu_rollout import _make, _weights_np

pytestmark = pytest.mark.gpu


def _bits_dense(bits_row):
    """(N,NW) int64 -> (N,N) bool"""
    b = bits_row.astype(np.uint64)
    N = b.shape[0]
    cols = np.arange(N)
    return ((b[:, cols >> 6] >> (cols & 63).astype(np.uint64)) & np.uint64(1)).astype(bool)


def _check_aggregate_and_its_update(mem, idx, X, G, Y, K, N, mean_pooling, hidden):
    """mgp_replay_aggregate against the dense gather of the same frames -- Z[s, f K + k] = X[s, k] . G[s, k] (reference
    actor.py:64-75), evaluated in fp64 from the gathered fp32 operands: 1e-6 of the row scale, zero taps exactly zero, labels
    identical -- and the update on it (mgp_train_grads_agg) against the update on (X, G) (mgp_train_grads): loss and every
    gradient entry to fp32 re-association of the aggregation."""
    import ctypes
    from multiagent_gnn_policies_amd import _lib, ops
    Bt = idx.numel()
    Z = torch.full((Bt, 6 * K, N), float('nan'), device='cuda')
    Yz = torch.full((Bt, 1, 2, N), float('nan'), device='cuda')
    ops.replay_aggregate(mem, idx, Z, Yz, mean_pooling)
    assert torch.equal(Yz, Y)
    ref = torch.einsum('skfm,skmn->sfkn', X.double(), G.double()).reshape(Bt, 6 * K, N)
    scale = ref.abs().amax(dim=2, keepdim=True).clamp_min(1.0)
    assert float(((Z.double() - ref).abs() / scale).max()) <= 1e-6
    ages = mem.age.view(-1)[idx]
    for k in range(K):
        young = ages < k
        if bool(young.any()):
            assert not bool(Z.view(Bt, 6, K, N)[young][:, :, k].any())
        assert torch.equal(Z.view(Bt, 6, K, N)[:, :, 0], X[:, 0])     # tap 0 is the frame's own feature block
    # two minibatches in one launch at a device-side cursor == single launches
    if Bt >= 12:
        idx2 = torch.cat([idx[:4], idx[7:11], idx[2:6]])
        Z2 = torch.empty((8, 6 * K, N), device='cuda'); Y2 = torch.empty((8, 1, 2, N), device='cuda')
        ops.replay_aggregate(mem, idx2, Z2, Y2, mean_pooling, cursor=torch.tensor([1], device='cuda', dtype=torch.int32), nb=2)
        sel = [7, 8, 9, 10, 2, 3, 4, 5]
        assert torch.equal(Z2, Z[sel]) and torch.equal(Y2, Y[sel])
    # the update
    L = _lib.lib()
    dims = (6,) + tuple(hidden) + (2,)
    cd = (ctypes.c_int * len(dims))(*dims)
    nl = len(dims) - 1
    Bu = min(Bt, 20)
    assert L.mgp_train_agg_supported(cd, nl, Bu, K, N)
    g = torch.Generator(device='cpu'); g.manual_seed(11)
    Ws, bs = [], []
    for l in range(nl):
        cin = dims[0] * K if l == 0 else dims[l]
        Ws.append((torch.randn((dims[l + 1], cin), generator=g) / np.sqrt(cin)).cuda())
        bs.append((0.1 * torch.randn((dims[l + 1],), generator=g)).cuda())
    P = sum(w.numel() + b_.numel() for w, b_ in zip(Ws, bs))
    wa = (ctypes.c_void_p * nl)(*[w.data_ptr() for w in Ws]); ba = (ctypes.c_void_p * nl)(*[b_.data_ptr() for b_ in bs])
    ws = torch.zeros((L.mgp_train_workspace(cd, nl, Bu, K, N),), device='cuda')
    ga, la = torch.zeros((P,), device='cuda'), torch.zeros((1,), device='cuda')
    _lib.check(L.mgp_train_grads_agg(ops._ptr(Z[:Bu].contiguous()), ops._ptr(Y[:Bu].contiguous()), wa, ba, cd, nl, ops._ptr(ga),
                                     ops._ptr(la), ops._ptr(ws), Bu, K, N, ops._stream()), 'mgp_train_grads_agg')
    if L.mgp_train_supported(cd, nl, Bu, K, N):
        gd, ld = torch.zeros((P,), device='cuda'), torch.zeros((1,), device='cuda')
        _lib.check(L.mgp_train_grads(ops._ptr(X[:Bu].contiguous()), ops._ptr(G[:Bu].contiguous()), ops._ptr(Y[:Bu].contiguous()), wa, ba,
                                     cd, nl, ops._ptr(gd), ops._ptr(ld), ops._ptr(ws), Bu, K, N, ops._stream()), 'mgp_train_grads')
        assert abs(float(la) - float(ld)) <= 1e-6 * max(1.0, abs(float(ld)))
        assert float((ga - gd).abs().max()) <= 1e-5 * max(1.0, float(gd.abs().max()))   # (measured: <= 6e-6 without mean pooling at K = 4, 5e-7 with)
    # and against autograd in fp64 on the aggregated input (reference gnn_dagger.py:85-93: mse_loss(actor(state), target))
    Wd = [w.double().requires_grad_(True) for w in Ws]; bd = [b_.double().requires_grad_(True) for b_ in bs]
    h = ref[:Bu]
    for l in range(nl):
        h = torch.einsum('oc,scn->son', Wd[l], h) + bd[l][None, :, None]
        if l < nl - 1:
            h = torch.tanh(h)
    loss = torch.nn.functional.mse_loss(h, Y[:Bu, 0].double())
    loss.backward()
    flat = torch.cat([torch.cat([w.grad.reshape(-1), b_.grad.reshape(-1)]) for w, b_ in zip(Wd, bd)])
    assert abs(float(la) - float(loss.detach())) <= 1e-5 * max(1.0, float(loss.detach()))
    assert float((ga.double() - flat).abs().max()) <= 1e-5 * max(1.0, float(flat.abs().max()))


def _setup(N, K, hidden, B, variant, ring_capacity):
    from multiagent_gnn_policies_amd.envs import VecFlock
    from multiagent_gnn_policies_amd.learner.vec_dagger import FrameReplay
    rs, op, actor, sim0, st = _make(N, K, hidden, B, seed=5, **variant)
    sim = VecFlock(B, sim0.p, 'cuda', with_expert=True)       # same states, with the expert by-product
    sim.set_state(sim0.x.cpu().numpy())
    st.reset(); st.push(sim.network, sim.features)
    mem = FrameReplay(B, ring_capacity, K, N, torch.device('cuda'))
    return op, actor, sim, st, mem


def _collect(actor, sim, st, mem, expert_io, beta, episode, seed, age0, T):
    from multiagent_gnn_policies_amd import ops
    from multiagent_gnn_policies_amd.learner.rollouts import _actor_params
    ws, bs = _actor_params(actor)
    flags = ops.RO_ENTER_CARRY | ops.RO_EXIT_CARRY | ops.RO_SKIP_DENSE
    assert st._carry_valid
    assert ops.rollout_collect(sim.x, st._G[st._cur], st.delay_state, tuple(actor.layers), sim._c, T, mem, expert_io, beta,
                               episode, seed, age0=age0, ring_step0=mem.head, carry=st.carry_buffer(), flags=flags,
                               weights=ws, biases=bs)
    mem.advance(T)
    st._pushes += T
    st._dense_stale = True


CASES = [(100, 3, (32, 32), {}), (50, 2, (16,), {'mean_pooling': False, 'n_leaders': 1}), (64, 4, (32, 32), {'centralized': False}),
         (100, 3, (32, 32), {'link_drop': 0.25, 'link_seed': 3}), (128, 1, (32,), {}),
         (200, 4, (32, 32), {'n_leaders': 2}), (150, 3, (32,), {'link_drop': 0.2, 'link_seed': 5})]     # rollout_big_kernel


@pytest.mark.parametrize('N,K,hidden,variant', CASES)
def test_collect_single_steps_match_oracle(N, K, hidden, variant):
    B, T, seed = 3, 7, 12345
    op, actor, sim, st, mem = _setup(N, K, hidden, B, variant, ring_capacity=B * T)
    Ws, bs = _weights_np(actor)
    beta_np = np.array([0.5, 0.8, 0.3], dtype=np.float32)
    beta = torch.from_numpy(beta_np).cuda()
    episode = torch.tensor([7, 1000, 123456], dtype=torch.int32, device='cuda')
    expert_io = sim.controller().permute(0, 2, 1).contiguous()
    n_expert = n_policy = 0
    for t in range(T):
        x0 = sim.x.cpu().numpy().copy()
        G0 = st.delay_gso.cpu().numpy().copy(); X0 = st.delay_state.cpu().numpy().copy()
        lab0 = expert_io.cpu().numpy().copy()
        ring_step = mem.head
        _collect(actor, sim, st, mem, expert_io, beta, episode, seed, age0=t, T=1)
        x1 = sim.x.cpu().numpy()
        pol = oa.forward(X0, G0, Ws, bs, 0, dtype=np.float64)               # (B,1,2,N)
        for b in range(B):
            # the frame of the state the step started from
            assert np.array_equal(mem.feat[ring_step, b].cpu().numpy(), X0[b, 0])
            assert int(mem.age[ring_step, b]) == t
            assert np.array_equal(mem.label[ring_step, b].cpu().numpy(), lab0[b])
            got_bits = _bits_dense(mem.bits[ring_step, b].cpu().numpy())
            want = (G0[b, 1] != 0) if (K > 1 and t >= 1) else np.zeros((N, N), dtype=bool)
            if K > 1:
                assert np.array_equal(got_bits, want), "membership bits of the frame's network"
            # its label is the expert's action for that state (FLOCK-SPEC section 5, p.centralized)
            lab_ref = ofl.controller(x0[b], op).T
            assert np.max(np.abs(lab0[b] - lab_ref) / np.maximum(1.0, np.abs(lab_ref))) <= 1e-6
            # who drove: the counter-based coin of oracle/dagger_vec.py
            drives = odv.expert_drives(seed, int(episode[b]), t, beta_np[b])
            x_exp = ofl.integrate(x0[b], lab0[b].T.astype(np.float32), op)
            x_pol = ofl.integrate(x0[b], pol[b, 0].T.astype(np.float32), op)
            if drives:
                assert np.array_equal(x1[b], x_exp), "expert-driven step: integration bit-exact given the stored label"
                n_expert += 1
            else:
                assert np.max(np.abs(x1[b] - x_pol)) <= 2e-6 and np.max(np.abs(x1[b] - x_exp)) > 1e-5
                n_policy += 1
            # and the transition: network of the new state bit-exact
            if K > 1:
                net = ofl.helpers(x1[b], op)['network'].astype(np.float32)
                assert np.array_equal(st.delay_gso[b, 1].cpu().numpy(), net)
    assert n_expert > 0 and n_policy > 0


@pytest.mark.parametrize('N,K,hidden,variant', CASES)
def test_collect_chunking_is_bit_identical_and_gather_rebuilds_the_states(N, K, hidden, variant):
    from multiagent_gnn_policies_amd import ops
    B, T, seed = 3, 9, 99
    beta = torch.tensor([0.5, 0.7, 0.2], device='cuda')
    episode = torch.tensor([3, 4, 5], dtype=torch.int32, device='cuda')
    runs = []
    for chunks in ([T], [1] * T, [4, 5]):
        op, actor, sim, st, mem = _setup(N, K, hidden, B, variant, ring_capacity=B * 6)     # ring shorter than the run: wraps
        expert_io = sim.controller().permute(0, 2, 1).contiguous()
        dense = []                                               # (X, G) of every visited state, from one-step launches
        t0 = 0
        for c in chunks:
            if c == 1:
                dense.append((st.delay_state.cpu().numpy().copy(), st.delay_gso.cpu().numpy().copy()))
            _collect(actor, sim, st, mem, expert_io, beta, episode, seed, age0=t0, T=c)
            t0 += c
        runs.append((sim.x.clone(), st.delay_state.clone(), st.delay_gso.clone(), expert_io.clone(), mem.feat.clone(),
                     mem.bits.clone(), mem.label.clone(), mem.age.clone(), mem, dense))
    for other in runs[1:]:
        for a, b_ in zip(runs[0][:8], other[:8]):
            assert torch.equal(a, b_)
    # rebuild every stored transition from the ring (window of 6 steps + K - 1 guard steps) and compare with the dense states
    mem, dense = runs[1][8], runs[1][9]
    assert mem.curr_size == B * 6 and mem.bytes_per_transition() == 32 * N + (16 if N <= 128 else 32) * N + 4
    ids = [mem.frame_of(i) for i in range(mem.curr_size)]
    idx = torch.tensor(ids, device='cuda', dtype=torch.long)
    Bt = len(ids)
    X = torch.empty((Bt, K, 6, N), device='cuda'); G = torch.empty((Bt, K, N, N), device='cuda')
    Y = torch.empty((Bt, 1, 2, N), device='cuda')
    ops.replay_gather(mem, idx, X, G, Y, op.mean_pooling)
    Xn, Gn = X.cpu().numpy(), G.cpu().numpy()
    for i in range(Bt):
        t = T - 6 + i // B                                       # positions run oldest to newest, lane-minor
        b = i % B
        Xd, Gd = dense[t]
        assert int(mem.age.view(-1)[ids[i]]) == t
        assert np.array_equal(Xn[i], Xd[b])
        assert np.array_equal(Gn[i, 0], np.eye(N, dtype=np.float32))
        if K > 1:
            assert np.array_equal(Gn[i, 1], Gd[b, 1])            # A_t itself: exact
        assert np.max(np.abs(Gn[i] - Gd[b])) <= 1e-6             # products: fp32 re-association only
    _check_aggregate_and_its_update(mem, idx, X, G, Y, K, N, op.mean_pooling, hidden)
    # device-side cursor form (what FrameUpdates replays): minibatch 1 of a (2, 4) index table
    tbl = torch.tensor([ids[:4], ids[5:9]], device='cuda', dtype=torch.long)
    X2 = torch.empty((4, K, 6, N), device='cuda'); G2 = torch.empty((4, K, N, N), device='cuda'); Y2 = torch.empty((4, 1, 2, N), device='cuda')
    ops.replay_gather(mem, tbl, X2, G2, Y2, op.mean_pooling, cursor=torch.ones((1,), device='cuda', dtype=torch.int32))
    assert torch.equal(X2, X[5:9]) and torch.equal(G2, G[5:9]) and torch.equal(Y2, Y[5:9])
    assert torch.equal(Y.view(Bt, 2, N), mem.label.view(-1, 2, N)[idx])


def test_collecting_build_two_episodes_per_cu_is_bit_identical(monkeypatch):
    """[r6] mgp_rollout_collect on 512-thread workgroups (csrc/rollout_t512.hip: selected when a launch has more lanes than the
    device has CUs, forced here with MGP_RO_T512): state, expert hand-over and every filed frame -- features, membership bits,
    labels, ages -- equal the 1024-thread build's bit for bit, one launch and chunked."""
    B, T, seed = 3, 23, 99
    beta = torch.tensor([0.5, 0.7, 0.2], device='cuda')
    episode = torch.tensor([3, 4, 5], dtype=torch.int32, device='cuda')
    runs = []
    for build, chunks in (('0', [T]), ('1', [T]), ('1', [1, 9, 13])):
        monkeypatch.setenv('MGP_RO_T512', build)
        op, actor, sim, st, mem = _setup(100, 3, (32, 32), B, {}, ring_capacity=B * 16)     # ring shorter than the run: wraps
        expert_io = sim.controller().permute(0, 2, 1).contiguous()
        t0 = 0
        for c in chunks:
            _collect(actor, sim, st, mem, expert_io, beta, episode, seed, age0=t0, T=c)
            t0 += c
        runs.append((sim.x.clone(), st.delay_state.clone(), st.delay_gso.clone(), expert_io.clone(), mem.feat.clone(),
                     mem.bits.clone(), mem.label.clone(), mem.age.clone()))
    for other in runs[1:]:
        for a, b_ in zip(runs[0], other):
            assert torch.equal(a, b_)


@pytest.mark.parametrize('aggregated', [True, False], ids=['aggregated', 'dense'])
def test_frame_updates_many_per_graph_equal_one_per_graph(aggregated):
    """A round of updates replayed 32-per-graph (device cursor, device step counter) is bit-identical to the same round replayed
    one update per graph, and to the eager path on gathered minibatches within fp32 rounding -- on the aggregated slots
    (mgp_replay_aggregate + mgp_train_step_agg, the default) and on the dense ones."""
    import configparser
    import random
    from multiagent_gnn_policies_amd.learner import vec_dagger as vd
    from multiagent_gnn_policies_amd.learner.gnn_dagger import DAGGER
    N, K, B, T = 100, 3, 8, 30
    op, actor, sim, st, mem = _setup(N, K, (32, 32), B, {}, ring_capacity=B * T)
    expert_io = sim.controller().permute(0, 2, 1).contiguous()
    _collect(actor, sim, st, mem, expert_io, torch.full((B,), 0.6, device='cuda'),
             torch.arange(B, dtype=torch.int32, device='cuda'), 7, age0=0, T=T)
    cp = configparser.ConfigParser()
    cp['DEFAULT'] = dict(n_states='6', n_actions='2', k=str(K), hidden_size='32', gamma='0.99', tau='0.5', n_agents=str(N),
                         actor_lr='1e-3')
    cp['t'] = {}
    U, Bt = 70, 20
    random.seed(4)
    ids = [mem.sample_ids(Bt) for _ in range(U)]
    outs = []
    for per_graph in (32, 10 ** 6, 0):
        torch.manual_seed(3)
        learner = DAGGER('cuda:0', cp['t'])
        if per_graph:
            vd.UPDATES_PER_GRAPH = per_graph if per_graph < 10 ** 6 else 32
            fu = vd.FrameUpdates(learner, mem, Bt, U, True, aggregated=aggregated)
            assert fu.aggregated == aggregated
            if per_graph == 10 ** 6:                             # one update per replay
                fu.idx[:U].copy_(torch.tensor(ids)); fu.cursor.zero_()
                torch.cuda.synchronize()
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    fu._enqueue()
                for _ in range(U):
                    g.replay()
                losses = fu.loss_hist[:U].clone()
            else:
                fu.run(ids)
                losses = fu.loss_hist[:U].clone()
                assert learner.actor_optim.step_count == U
            assert int(fu.cursor.item()) == U and int(learner.actor_optim.step_dev.item()) == U
        else:                                                    # eager: gather, then the learner's own update
            X = torch.empty((Bt, K, 6, N), device='cuda'); G = torch.empty((Bt, K, N, N), device='cuda')
            Y = torch.empty((Bt, 1, 2, N), device='cuda')
            ls = []
            for u in range(U):
                from multiagent_gnn_policies_amd import ops
                ops.replay_gather(mem, torch.tensor(ids[u], device='cuda'), X, G, Y, True)
                ls.append(learner.gradient_step_tensors(X, G, Y))
            losses = torch.tensor(ls, device='cuda')
        outs.append((losses.cpu().numpy(), learner.actor_optim.flat.clone().cpu().numpy()))
    vd.UPDATES_PER_GRAPH = 32
    assert np.array_equal(outs[0][0], outs[1][0]) and np.array_equal(outs[0][1], outs[1][1])
    assert np.max(np.abs(outs[0][0] - outs[2][0])) <= 1e-5 and np.max(np.abs(outs[0][1] - outs[2][1])) <= 1e-5
    assert outs[0][0][-1] < outs[0][0][0]                         # and it learns


# ---------------------------------------------------------------------------------------------------------------------
# N > 256: the same collection on the factored state in HBM (mgp_sparse_policy_collect, K launches per env step) and the
# row-wise gather of its frames (mgp_replay_gather_rows)
SPARSE_CASES = [(300, 3, (32, 32), {}), (1000, 3, (32, 32), {}), (300, 4, (32,), {'mean_pooling': False, 'n_leaders': 2}),
                (320, 2, (16, 16), {'link_drop': 0.2, 'link_seed': 5}), (260, 1, (32,), {'centralized': False})]


def _setup_sparse(N, K, hidden, B, variant, ring_capacity):
    from multiagent_gnn_policies_amd.envs import VecFlock
    from multiagent_gnn_policies_amd.learner.sparse_rollout import SparseFlockState
    from multiagent_gnn_policies_amd.learner.vec_dagger import FrameReplay
    rs, op, actor, sim0, st = _make(N, K, hidden, B, seed=5, **variant)
    sim = VecFlock(B, sim0.p, 'cuda', with_expert=True)
    sim.set_state(sim0.x.cpu().numpy())
    st.reset(); st.push(sim.network, sim.features)
    mem = FrameReplay(B, ring_capacity, K, N, torch.device('cuda'))
    assert mem.wrow is not None and mem.nw % 8 == 0 and mem.nw * 64 >= N
    sp = SparseFlockState(sim, K)
    sp.observe_reset(sim)
    return op, actor, sim, st, mem, sp


@pytest.mark.parametrize('N,K,hidden,variant', SPARSE_CASES)
def test_sparse_collect_single_steps_match_oracle(N, K, hidden, variant):
    """Reference gnn_dagger.py:154-178 per lane, every step against the oracle: the frame (features, membership bits, row
    weights, age), the label (FLOCK-SPEC section 5 on the state before the step), the coin (oracle/dagger_vec.py), the
    expert-driven step bit-exact given the stored label, the policy-driven step against the oracle forward on the dense
    state, and the network of the new state bit-exact."""
    from multiagent_gnn_policies_amd.learner.sparse_rollout import sparse_collect
    B, T, seed = 3, 6, 4321
    op, actor, sim, st, mem, sp = _setup_sparse(N, K, hidden, B, variant, ring_capacity=B * T)
    Ws, bs = _weights_np(actor)
    beta_np = np.array([0.5, 0.8, 0.3], dtype=np.float32)
    beta = torch.from_numpy(beta_np).cuda()
    episode = torch.tensor([7, 1000, 123456], dtype=torch.int32, device='cuda')
    n_expert = n_policy = 0
    for t in range(T):
        sp.to_dense(sim, st)
        x0 = sim.x.cpu().numpy().copy()
        G0 = st.delay_gso.cpu().numpy().copy(); X0 = st.delay_state.cpu().numpy().copy()
        lab0 = sim.expert.cpu().numpy().copy()                               # (B,N,2)
        ring_step = mem.head
        sparse_collect(actor, sim, sp, mem, beta, episode, seed, t, 1)
        assert mem.head == (ring_step + 1) % mem.ring_steps
        x1 = sim.x.cpu().numpy()
        pol = oa.forward(X0, G0, Ws, bs, 0, dtype=np.float64)               # (B,1,2,N)
        for b in range(B):
            h0 = ofl.helpers(x0[b], op)
            assert np.array_equal(mem.feat[ring_step, b].cpu().numpy(), X0[b, 0])
            assert int(mem.age[ring_step, b]) == t
            assert np.array_equal(mem.label[ring_step, b].cpu().numpy(), lab0[b].T)
            got_bits = _bits_dense(mem.bits[ring_step, b].cpu().numpy())
            assert np.array_equal(got_bits, h0['network'] != 0), "membership bits of the frame's network"
            w_ref = h0['network'].astype(np.float32).max(axis=1)
            w_ref = np.where(w_ref == 0, np.float32(1.0), w_ref)             # isolated agents: weight 1 (1 / max(deg, 1))
            assert np.array_equal(mem.wrow[ring_step, b].cpu().numpy(), w_ref)
            lab_ref = ofl.controller(x0[b], op)
            assert np.max(np.abs(lab0[b] - lab_ref) / np.maximum(1.0, np.abs(lab_ref))) <= 1e-6
            drives = odv.expert_drives(seed, int(episode[b]), t, beta_np[b])
            x_exp = ofl.integrate(x0[b], lab0[b].astype(np.float32), op)
            x_pol = ofl.integrate(x0[b], pol[b, 0].T.astype(np.float32), op)
            if drives:
                assert np.array_equal(x1[b], x_exp), "expert-driven step: integration bit-exact given the stored label"
                n_expert += 1
            else:
                assert np.max(np.abs(x1[b] - x_pol)) <= 2e-6 and np.max(np.abs(x1[b] - x_exp)) > 1e-5
                n_policy += 1
            net = ofl.helpers(x1[b], op)['network'].astype(np.float32)
            from test_gpu_sparse import _bits_to_dense
            got = _bits_to_dense(sp.bits[b, sp.hs].cpu().numpy(), sp.wrow[b, sp.hs].cpu().numpy(), N)
            assert np.array_equal(got, net)
    assert n_expert > 0 and n_policy > 0


@pytest.mark.parametrize('N,K,hidden,variant', SPARSE_CASES)              # incl. configs[2]'s N = 1000 at full size, and K = 1
def test_sparse_collect_gather_rebuilds_the_states(N, K, hidden, variant):
    """One call of T steps equals T one-step calls bit for bit (frames, state), and every stored transition's K-tap state
    rebuilt from the ring (window shorter than the run: wraps; K - 1 guard steps) equals the dense state the factored path
    materialises at that step: delay line and A_t exact, products 1e-6 (state_with_delay.py:44-53 on the stored history)."""
    from multiagent_gnn_policies_amd import ops
    from multiagent_gnn_policies_amd.learner.sparse_rollout import sparse_collect
    B, T, seed = 3, 9, 99
    beta = torch.tensor([0.5, 0.7, 0.2], device='cuda')
    episode = torch.tensor([3, 4, 5], dtype=torch.int32, device='cuda')
    runs = []
    for chunks in ([T], [1] * T):
        op, actor, sim, st, mem, sp = _setup_sparse(N, K, hidden, B, variant, ring_capacity=B * 6)
        dense = []
        t0 = 0
        for c in chunks:
            if c == 1:
                sp.to_dense(sim, st)
                dense.append((st.delay_state.cpu().numpy().copy(), st.delay_gso.cpu().numpy().copy()))
            sparse_collect(actor, sim, sp, mem, beta, episode, seed, t0, c)
            t0 += c
        runs.append((sim.x.clone(), sp.bits.clone(), sp.wrow.clone(), sp.feat.clone(), sim.expert.clone(), mem.feat.clone(),
                     mem.bits.clone(), mem.wrow.clone(), mem.label.clone(), mem.age.clone(), mem, dense))
    for a, b_ in zip(runs[0][:10], runs[1][:10]):
        assert torch.equal(a, b_)
    mem, dense = runs[1][10], runs[1][11]
    assert mem.curr_size == B * 6 and mem.bytes_per_transition() == 32 * N + 8 * mem.nw * N + 4 * N + 4
    ids = [mem.frame_of(i) for i in range(mem.curr_size)]
    idx = torch.tensor(ids, device='cuda', dtype=torch.long)
    Bt = len(ids)
    X = torch.empty((Bt, K, 6, N), device='cuda'); G = torch.empty((Bt, K, N, N), device='cuda')
    Y = torch.empty((Bt, 1, 2, N), device='cuda')
    ops.replay_gather(mem, idx, X, G, Y, op.mean_pooling)
    Xn, Gn = X.cpu().numpy(), G.cpu().numpy()
    for i in range(Bt):
        t = T - 6 + i // B
        b = i % B
        Xd, Gd = dense[t]
        assert int(mem.age.view(-1)[ids[i]]) == t
        assert np.array_equal(Xn[i], Xd[b])
        assert np.array_equal(Gn[i, 0], np.eye(N, dtype=np.float32))
        if K > 1:
            assert np.array_equal(Gn[i, 1], Gd[b, 1])            # A_t itself: exact
        for j in range(2, K):
            assert np.max(np.abs(Gn[i, j] - Gd[b, j])) <= 1e-6 * max(1.0, float(np.max(np.abs(Gd[b, j]))))
            if t < j:
                assert not Gn[i, j].any()
    _check_aggregate_and_its_update(mem, idx, X, G, Y, K, N, op.mean_pooling, hidden)
    # two minibatches in one launch at a device-side cursor == two single gathers
    idx2 = torch.tensor(ids[:4] + ids[7:11] + ids[2:6], device='cuda', dtype=torch.long)
    cursor = torch.tensor([1], device='cuda', dtype=torch.int32)
    X2 = torch.empty((8, K, 6, N), device='cuda'); G2 = torch.empty((8, K, N, N), device='cuda'); Y2 = torch.empty((8, 1, 2, N), device='cuda')
    ops.replay_gather(mem, idx2, X2, G2, Y2, op.mean_pooling, cursor=cursor, nb=2)
    sel = [7, 8, 9, 10, 2, 3, 4, 5]
    assert torch.equal(X2, X[sel]) and torch.equal(G2, G[sel]) and torch.equal(Y2, Y[sel])


def test_vectorised_dagger_collects_on_the_factored_state():
    """train_dagger_vec at N = 300 (beyond the resident kernel): the rounds collect on the device (no host-stepped loop, no
    dense replay), the updates run from the gathered frames, the losses are finite and the policy evaluation runs."""
    import configparser
    from multiagent_gnn_policies_amd.learner.vec_dagger import train_dagger_vec
    cp = configparser.ConfigParser()
    cp['DEFAULT'] = dict(env='FlockingRelative-v0', alg='dagger_vec', n_states='6', n_actions='2', k='3', hidden_size='32',
                         gamma='0.99', tau='0.5', n_agents='300', comm_radius='1.0', v_max='3.0', actor_lr='5e-5',
                         buffer_size='400', batch_size='8', updates_per_step='3', n_train_episodes='8', n_test_episodes='4',
                         beta_coeff='0.993', seed='3', debug='False')
    cp['t'] = {}
    np.random.seed(3); torch.manual_seed(3)
    import random
    random.seed(3)
    res = train_dagger_vec(cp['t'], 'cuda', n_envs=4, episode_steps=30)
    assert res['collect'] == 'device' and res['updates'] == 2 * 3 * 4
    assert res['replay_bytes_per_transition'] == 32 * 300 + 8 * 8 * 300 + 4 * 300 + 4
    assert np.isfinite(res['mean']) and res['mean'] < 0


@pytest.mark.parametrize('N,hidden,variant', [(300, (32, 32), {}), (1000, (32, 32), {}), (600, (32,), {'mean_pooling': False, 'centralized': False})])
def test_sparse_collect_persistent_form_equals_the_k_launch_form(N, hidden, variant, monkeypatch):
    """DAGGER collection on the factored state at K = 3 runs inside the persistent launch (csrc/sparse_persist.hip: frames
    filed and the expert's action substituted in its policy phase): frames, labels, ages, the state and every ring equal the
    K-launch form's (mgp_sparse_policy_collect per step) bit for bit."""
    from multiagent_gnn_policies_amd.learner.sparse_rollout import sparse_collect
    K, B, T, seed = 3, 3, 8, 77
    beta = torch.tensor([0.5, 0.9, 0.1], device='cuda')
    episode = torch.tensor([11, 12, 99], dtype=torch.int32, device='cuda')
    runs = []
    for persist in ('0', '1'):
        monkeypatch.setenv('MGP_SP_PERSIST', persist)
        op, actor, sim, st, mem, sp = _setup_sparse(N, K, hidden, B, variant, ring_capacity=B * 5)
        rewards = torch.zeros((B, T), device='cuda', dtype=torch.float64)
        sparse_collect(actor, sim, sp, mem, beta, episode, seed, 0, 3, rewards=rewards[:, :3].clone())
        sparse_collect(actor, sim, sp, mem, beta, episode, seed, 3, T - 3)
        sp.check_status()
        runs.append((sim.x.clone(), sp.bits.clone(), sp.wrow.clone(), sp.feat.clone(), sp.nbr.clone(), sim.expert.clone(), mem.feat.clone(),
                     mem.bits.clone(), mem.wrow.clone(), mem.label.clone(), mem.age.clone()))
    for a, b_ in zip(*runs):
        assert torch.equal(a, b_)
    assert runs[0][6].abs().sum() > 0 and runs[0][7].any()
