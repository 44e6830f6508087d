"""GPU parity: every HIP kernel (through the C ABI via ops.py) against the numpy oracle on the same
seeded inputs.  fp32 tolerance 1e-5 relative to max(1,|ref|) (north star: 1e-5 fp32)."""
import numpy as np
import pytest
import torch

from oracle import actor as oa, state as os_, dagger as od, flock as ofl, synth

pytestmark = pytest.mark.gpu

TOL = 1e-5


def dev(a, dtype=torch.float32):
    return torch.from_numpy(np.ascontiguousarray(a)).to('cuda', dtype=dtype)


def relerr(a, b):
    a = np.asarray(a, dtype=np.float64); b = np.asarray(b, dtype=np.float64)
    return float(np.max(np.abs(a - b) / np.maximum(1.0, np.abs(b)))) if a.size else 0.0


AGG_SHAPES = [(1, 3, 6, 100), (4, 3, 6, 100), (2, 4, 6, 200), (2, 2, 6, 16), (2, 1, 6, 16), (3, 2, 3, 7),
              (2, 3, 6, 33), (1, 3, 6, 260), (1, 2, 6, 1000), (2, 3, 32, 100), (1, 2, 40, 36), (2, 3, 1, 64),
              (1, 2, 12, 257), (1, 1, 6, 513), (2, 2, 8, 128), (1, 3, 16, 48),
              (2, 2, 6, 129), (2, 2, 6, 130), (1, 3, 6, 255), (1, 2, 3, 201),     # V = 1, one row phase (fuzz-found bug)
              # the MFMA variant's corners (16 <= N <= 128, N % 4 == 0, C <= 8, K * column blocks <= 8) and just outside
              (2, 4, 6, 64), (1, 3, 5, 68), (2, 3, 3, 124), (3, 4, 8, 128), (1, 8, 2, 20), (1, 5, 6, 100), (2, 3, 9, 100)]


@pytest.mark.parametrize('shape', AGG_SHAPES)
@pytest.mark.parametrize('dense', [False, True])
def test_agg_fwd(shape, dense):
    from multiagent_gnn_policies_amd import ops
    B, K, C, N = shape
    X, G = (synth.make_dense_inputs if dense else synth.make_inputs)(5, B, K, C, N)
    ref = oa.aggregate_bkfn(X.astype(np.float64), G.astype(np.float64))          # (B,K,C,N)
    # input in the (B,K,C,N) layout viewed as (B,C,K,N) -- the ind_agg == 0 case
    T = dev(X).permute(0, 2, 1, 3)
    Y = ops.agg_fwd(T, dev(G))
    assert Y.shape == (B, C, K, N)
    assert relerr(Y.permute(0, 2, 1, 3).cpu().numpy(), ref) <= TOL
    # input already (B,C,K,N) contiguous -- the ind_agg > 0 case
    T2 = dev(np.transpose(X, (0, 2, 1, 3)))
    Y2 = ops.agg_fwd(T2, dev(G))
    assert relerr(Y2.permute(0, 2, 1, 3).cpu().numpy(), ref) <= TOL


def test_agg_fwd_identity_and_asymmetric():
    """G = I returns X; a strictly asymmetric G catches a transposed contraction."""
    from multiagent_gnn_policies_amd import ops
    B, K, C, N = 2, 2, 6, 100
    rs = np.random.RandomState(0)
    X = rs.randn(B, K, C, N).astype(np.float32)
    G = np.zeros((B, K, N, N), np.float32)
    G[:, 0] = np.eye(N)
    G[:, 1] = np.triu(rs.rand(N, N), 1)          # upper triangular: G != G^T
    Y = ops.agg_fwd(dev(X).permute(0, 2, 1, 3), dev(G)).permute(0, 2, 1, 3).cpu().numpy()
    assert np.array_equal(Y[:, 0], X[:, 0])
    assert relerr(Y[:, 1], np.einsum('bcm,bmn->bcn', X[:, 1].astype(np.float64), G[:, 1].astype(np.float64))) <= TOL
    assert np.abs(Y[:, 1, :, 0]).max() == 0.0    # column 0 of a strictly upper-triangular G is empty


@pytest.mark.parametrize('shape', [(2, 3, 6, 100), (1, 2, 3, 7), (2, 3, 32, 33), (1, 2, 6, 260), (2, 1, 16, 64)])
def test_agg_bwd_x(shape):
    from multiagent_gnn_policies_amd import ops
    B, K, C, N = shape
    _, G = synth.make_dense_inputs(9, B, K, C, N)
    dY = np.random.RandomState(1).randn(B, C, K, N).astype(np.float32)
    ref = np.einsum('bckn,bkmn->bckm', dY.astype(np.float64), G.astype(np.float64))
    got = ops.agg_bwd_x(dev(dY), dev(G)).cpu().numpy()
    assert relerr(got, ref) <= TOL


@pytest.mark.parametrize('cfg', [(2, 18, 32, 1, 100, 1), (3, 32, 32, 1, 100, 1), (2, 32, 2, 1, 100, 0),
                                 (2, 6, 16, 3, 16, 1), (1, 5, 3, 2, 7, 1), (2, 128, 128, 1, 33, 1),
                                 (1, 200, 70, 1, 65, 0), (2, 4, 1, 1, 130, 0)])
def test_dense_fwd_bwd(cfg):
    from multiagent_gnn_policies_amd import ops
    B, Cin, Cout, T, N, act = cfg
    rs = np.random.RandomState(3)
    x = rs.randn(B, Cin, T, N).astype(np.float32)
    W = (rs.randn(Cout, Cin) / np.sqrt(Cin)).astype(np.float32)
    b = rs.randn(Cout).astype(np.float32)
    z = np.einsum('oc,bctn->botn', W.astype(np.float64), x.astype(np.float64)) + b[None, :, None, None]
    ref = np.tanh(z) if act else z
    xt, Wt, bt = dev(x).requires_grad_(True), dev(W).requires_grad_(True), dev(b).requires_grad_(True)
    out = ops.dense(xt, Wt, bt, act)
    assert relerr(out.detach().cpu().numpy(), ref) <= TOL
    dO = rs.randn(*ref.shape).astype(np.float32)
    out.backward(dev(dO))
    delta = dO.astype(np.float64) * ((1 - ref * ref) if act else 1.0)
    assert relerr(Wt.grad.cpu().numpy(), np.einsum('botn,bctn->oc', delta, x.astype(np.float64))) <= 2e-5
    assert relerr(bt.grad.cpu().numpy(), delta.sum(axis=(0, 2, 3))) <= 2e-5
    assert relerr(xt.grad.cpu().numpy(), np.einsum('oc,botn->bctn', W.astype(np.float64), delta)) <= 2e-5


def test_dense_strided_input():
    """delay_state (B,K,F,N) read through a permuted view (first layer when ind_agg > 0)."""
    from multiagent_gnn_policies_amd import ops
    rs = np.random.RandomState(4)
    X = rs.randn(2, 3, 6, 50).astype(np.float32)
    W = rs.randn(8, 6).astype(np.float32); b = rs.randn(8).astype(np.float32)
    out = ops.dense_fwd(dev(X).permute(0, 2, 1, 3), dev(W), dev(b), 1).cpu().numpy()
    ref = np.tanh(np.einsum('oc,bkcn->bokn', W.astype(np.float64), X.astype(np.float64)) + b[None, :, None, None])
    assert relerr(out, ref) <= TOL


@pytest.mark.parametrize('cfg', [(3, 3, 6, 100), (2, 1, 6, 16), (2, 2, 6, 16), (2, 4, 6, 200), (1, 3, 6, 33),
                                 (1, 5, 4, 7), (1, 3, 6, 1000)])
@pytest.mark.parametrize('dense_a', [False, True])
def test_gso_update(cfg, dense_a):
    from multiagent_gnn_policies_amd import ops
    B, K, F, N = cfg
    rs = np.random.RandomState(11)
    Gp = Xp = None
    Gp_d = Xp_d = None
    for t in range(K + 1):
        if dense_a:
            A = (rs.rand(B, N, N) * (rs.rand(B, N, N) < 0.5)).astype(np.float32)
            for b in range(B):
                np.fill_diagonal(A[b], 0.0)
        else:
            A = synth.make_adjacency_batch(100 + t, B, N)
        X = rs.randn(B, F, N).astype(np.float32)
        Gn, Xn = os_.gso_update(A.astype(np.float64), Gp, X, Xp, K, dtype=np.float64)
        Gd, Xd = ops.gso_update(dev(A), Gp_d, dev(X), Xp_d, K)
        assert relerr(Gd.cpu().numpy(), Gn) <= TOL
        assert np.array_equal(Xd.cpu().numpy(), Xn.astype(np.float32))
        assert np.array_equal(Gd[:, 0].cpu().numpy(), np.broadcast_to(np.eye(N, dtype=np.float32), (B, N, N)))
        if t > 0 and K > 1:
            assert np.array_equal(Gd[:, 1].cpu().numpy(), A)          # A @ I is exact (state_with_delay.py:47)
        Gp, Xp = Gn, Xn
        Gp_d, Xp_d = Gd, Xd


@pytest.mark.parametrize('cfg', [(2, 3, 100), (1, 4, 16), (2, 1, 16), (1, 2, 33)])
def test_gso_powers(cfg):
    from multiagent_gnn_policies_amd import ops
    B, K, N = cfg
    A = synth.make_adjacency_batch(7, B, N)
    ref = os_.gso_powers(A.astype(np.float64), K, dtype=np.float64)
    assert relerr(ops.gso_powers(dev(A), K).cpu().numpy(), ref) <= TOL


def _flock_params(n, **kw):
    return ofl.FlockParams(n_agents=n, **kw)


def _c_params(p):
    from multiagent_gnn_policies_amd.envs import FlockParams
    return FlockParams(**{f: getattr(p, f) for f in FlockParams.__dataclass_fields__}).to_c()


@pytest.mark.parametrize('n', [100, 16, 33, 200, 1000])
@pytest.mark.parametrize('variant', [{}, {'mean_pooling': False}, {'n_leaders': 3}, {'centralized': False},
                                     {'link_drop': 0.3, 'link_seed': 7},
                                     {'comm_radius': 4.0}])      # rad.cfg's largest radius: near-complete graphs
def test_flock_step_and_controller(n, variant):
    from multiagent_gnn_policies_amd import ops
    p = _flock_params(n, **variant)
    rs = np.random.RandomState(n)
    B = 3 if n <= 200 else 1
    xs = np.stack([ofl.sample_candidate(rs, p) for _ in range(B)])
    us = rs.uniform(-1.5, 1.5, size=(B, n, 2)).astype(np.float32)        # beyond the clip on purpose
    x_d = dev(xs, torch.float64)
    A = torch.empty((B, n, n), device='cuda'); A64 = torch.empty((B, n, n), device='cuda', dtype=torch.float64)
    feat = torch.empty((B, 6, n), device='cuda'); feat64 = torch.empty((B, n, 6), device='cuda', dtype=torch.float64)
    rew = torch.empty((B,), device='cuda', dtype=torch.float64)
    cp = _c_params(p)
    for it in range(3):
        ex_d = torch.empty((B, n, 2), device='cuda')
        ops.flock_step(x_d, dev(us), cp, A=A, A64=A64, feat=feat, feat64=feat64, reward=rew, expert=ex_d)
        u_d = torch.empty((B, n, 2), device='cuda'); u64_d = torch.empty((B, n, 2), device='cuda', dtype=torch.float64)
        ops.flock_controller(x_d, cp, centralized=False, u=u_d, u64=u64_d)
        uc_d = torch.empty((B, n, 2), device='cuda', dtype=torch.float64)
        ops.flock_controller(x_d, cp, centralized=True, u64=uc_d)
        ub_d = torch.empty((B, n, 2), device='cuda')
        ops.flock_controller(x_d, cp, centralized=p.centralized, u=ub_d)
        for b in range(B):
            x2, vals, net, r = ofl.step(xs[b], us[b], p)
            assert np.array_equal(x_d[b].cpu().numpy(), x2), "integration must be bit-exact fp64"
            assert np.array_equal(A64[b].cpu().numpy(), net), "adjacency must be bit-exact"
            assert np.array_equal(A[b].cpu().numpy(), net.astype(np.float32))
            assert np.sum(np.diag(A64[b].cpu().numpy())) == 0
            assert relerr(feat64[b].cpu().numpy(), vals) <= 1e-11
            assert relerr(feat[b].cpu().numpy(), vals.T.astype(np.float32)) <= 1e-6
            assert abs(rew[b].item() - r) <= 1e-12 * max(1.0, abs(r))
            uo = ofl.controller(x2, p, centralized=False)
            assert relerr(u64_d[b].cpu().numpy(), uo) <= 1e-11
            assert relerr(u_d[b].cpu().numpy(), uo) <= 1e-6
            assert np.array_equal(ex_d[b].cpu().numpy(), ub_d[b].cpu().numpy())    # by-product == dedicated call
            assert relerr(uc_d[b].cpu().numpy(), ofl.controller(x2, p, centralized=True)) <= 1e-11
            xs[b] = x2
        us = rs.uniform(-1.0, 1.0, size=(B, n, 2)).astype(np.float32)


def test_flock_refresh_without_action():
    from multiagent_gnn_policies_amd import ops
    p = _flock_params(50)
    xs = ofl.sample_candidate(np.random.RandomState(1), p)[None]
    x_d = dev(xs, torch.float64)
    A64 = torch.empty((1, 50, 50), device='cuda', dtype=torch.float64)
    ops.flock_step(x_d, None, _c_params(p), A64=A64)
    assert np.array_equal(x_d.cpu().numpy(), xs)
    assert np.array_equal(A64[0].cpu().numpy(), ofl.helpers(xs[0], p)['network'])


def test_mse_and_adam():
    from multiagent_gnn_policies_amd import ops
    rs = np.random.RandomState(0)
    pred = rs.randn(20, 1, 2, 100).astype(np.float32); tgt = rs.randn(20, 1, 2, 100).astype(np.float32)
    loss, g = ops.mse_grad(dev(pred), dev(tgt))
    assert abs(loss.item() - od.mse_loss(pred, tgt)) <= 1e-6
    assert relerr(g.cpu().numpy(), od.mse_grad(pred, tgt)) <= 1e-6
    n = 1730
    p = rs.randn(n).astype(np.float32); m = np.zeros(n, np.float32); v = np.zeros(n, np.float32)
    p_d, m_d, v_d = dev(p), dev(m), dev(v)
    P, M, V = [p], [m], [v]
    for step in range(1, 6):
        grad = (rs.randn(n) * 10 ** rs.uniform(-4, 1, size=n)).astype(np.float32)
        ops.adam_step(p_d, dev(grad), m_d, v_d, 5e-5, step)
        P, M, V = od.adam_step(P, [grad], M, V, step, 5e-5)
        assert np.max(np.abs(p_d.cpu().numpy() - P[0])) <= 2e-7


@pytest.mark.parametrize('cfg', [(3, 3, 6, 100), (2, 4, 6, 36), (2, 2, 6, 16), (2, 1, 6, 16)])
def test_inplace_state_protocol_equals_classic_update(cfg):
    """sim -> strided slots -> mgp_gso_advance reproduces mgp_gso_update bit-for-bit, including the first step."""
    from multiagent_gnn_policies_amd.envs import FlockParams, VecFlock
    from multiagent_gnn_policies_amd.learner.state_with_delay import BatchedDelayState
    B, K, F, N = cfg
    p = FlockParams(n_agents=N, init_mode='grid')
    sims = [VecFlock(B, p, 'cuda') for _ in range(2)]
    x0 = np.stack([ofl.reset(np.random.RandomState(5 + b), ofl.FlockParams(n_agents=N, init_mode='grid')) for b in range(B)])
    for s in sims:
        s.set_state(x0)
    classic, inplace = BatchedDelayState('cuda', B, K, F, N), BatchedDelayState('cuda', B, K, F, N)
    classic.push(sims[0].network, sims[0].features)
    A_dst, X_dst = inplace.next_slots()
    sims[1].step(None, A_out=A_dst, feat_out=X_dst)
    inplace.advance()
    rs = np.random.RandomState(0)
    for t in range(K + 2):
        assert torch.equal(classic.delay_gso, inplace.delay_gso)
        assert torch.equal(classic.delay_state, inplace.delay_state)
        u = dev(rs.uniform(-1, 1, size=(B, N, 2)).astype(np.float32))
        sims[0].step(u)
        classic.push(sims[0].network, sims[0].features)
        A_dst, X_dst = inplace.next_slots()
        sims[1].step(u, A_out=A_dst, feat_out=X_dst)
        inplace.advance()
    inplace.reset(); classic.reset()                         # new episode: taps >= 1 must read zero again
    classic.push(sims[0].network, sims[0].features)
    A_dst, X_dst = inplace.next_slots()
    sims[1].step(None, A_out=A_dst, feat_out=X_dst)
    inplace.advance()
    assert torch.equal(classic.delay_gso, inplace.delay_gso) and torch.equal(classic.delay_state, inplace.delay_state)
    if K > 1:
        assert float(inplace.delay_gso[:, 1:].abs().max()) == 0.0


@pytest.mark.parametrize('n', [100, 33, 200, 1000])
def test_flock_pingpong_equals_inplace(n):
    """x_out != x (several workgroups per episode, redundant integration) is bit-identical to the in-place form."""
    from multiagent_gnn_policies_amd import ops
    p = _flock_params(n, init_mode='grid')
    B = 2
    rs = np.random.RandomState(n)
    xs = np.stack([ofl.reset(rs, p) for _ in range(B)])
    us = dev(rs.uniform(-1.2, 1.2, size=(B, n, 2)).astype(np.float32))
    cp = _c_params(p)
    outs = []
    for pingpong in (False, True):
        x = dev(xs, torch.float64)
        xo = torch.zeros_like(x) if pingpong else None
        A = torch.empty((B, n, n), device='cuda'); feat = torch.empty((B, 6, n), device='cuda')
        rew = torch.empty((B,), device='cuda', dtype=torch.float64); ex = torch.empty((B, n, 2), device='cuda')
        ops.flock_step(x, us, cp, A=A, feat=feat, reward=rew, expert=ex, x_out=xo)
        if pingpong:
            assert np.array_equal(x.cpu().numpy(), xs)                    # the source state is left untouched
        outs.append((xo if pingpong else x, A, feat, rew, ex))
    for a, b_ in zip(outs[0], outs[1]):
        assert torch.equal(a, b_)


@pytest.mark.parametrize('cfg', [(3, 3, 100), (2, 4, 36), (2, 2, 16), (2, 3, 128), (2, 5, 128), (2, 6, 128), (2, 7, 100)])
def test_fused_sim_state_kernel_equals_two_kernel_protocol(cfg):
    """mgp_flock_step_advance == mgp_flock_step (strided outputs) + mgp_gso_advance, bit for bit, incl. episode starts.
    (K - 1) 6 N beyond 3072 -- K = 6 at N = 128, K = 7 at N = 100 -- is past the one-workgroup kernel's delay-line registers: the
    dispatch must hand those shapes to the row-tiled kernel (every tap of Xd_next written)."""
    from multiagent_gnn_policies_amd.envs import FlockParams, VecFlock
    from multiagent_gnn_policies_amd.learner.state_with_delay import BatchedDelayState
    B, K, N = cfg
    p = FlockParams(n_agents=N, init_mode='grid')
    x0 = np.stack([ofl.reset(np.random.RandomState(9 + b), ofl.FlockParams(n_agents=N, init_mode='grid')) for b in range(B)])
    sims = [VecFlock(B, p, 'cuda', with_expert=True) for _ in range(2)]
    states = [BatchedDelayState('cuda', B, K, 6, N) for _ in range(2)]
    rs = np.random.RandomState(1)
    for epi in range(2):
        for s_, st in zip(sims, states):
            s_.set_state(x0 + 0.01 * epi)
            st.reset()
            st.push(s_.network, s_.features)
        for t in range(K + 2):
            u = dev(rs.uniform(-1.1, 1.1, size=(B, 1, 2, N)).astype(np.float32))
            A_dst, X_dst = states[0].next_slots()
            sims[0].step(u, A_out=A_dst, feat_out=X_dst)
            states[0].advance()
            sims[1].step_advance(u, states[1])
            assert torch.equal(sims[0].x, sims[1].x)
            assert torch.equal(states[0].delay_gso, states[1].delay_gso)
            assert torch.equal(states[0].delay_state, states[1].delay_state)
            assert torch.equal(sims[0].reward, sims[1].reward) and torch.equal(sims[0].expert, sims[1].expert)
    # episode start through the fused kernel: taps >= 1 read zero
    st = BatchedDelayState('cuda', B, K, 6, N)
    sims[1].step_advance(dev(np.zeros((B, 1, 2, N), np.float32)), st)
    assert float(st.delay_gso[:, 1:].abs().max()) == 0.0 and float(st.delay_state[:, 1:].abs().max()) == 0.0
    assert torch.equal(st.delay_gso[:, 0], torch.eye(N, device='cuda').expand(B, N, N))


@pytest.mark.parametrize('hidden', [(8,), (32, 32)])
def test_tanh_fast_absolute_error(hidden):
    """The hidden layers evaluate tanh as 1 - 2 / (1 + exp2(2 x log2 e)) on v_exp_f32 / v_rcp_f32 (mgp_device.h::tanh_fast: five
    instructions, no branches) -- the approximation SURVEY section 7 warns about.  With identity-like weights the fused Actor
    forward returns tanh_fast(x) itself (one hidden layer of 8: the generic fp32-MFMA chain) or tanh_fast(tanh_fast(x)) (the
    reference's policy shape [32, 32]: actor_fwd_pol_kernel, whose layers are the resident kernel's ro_layer_bf16 -- a unit
    weight times x is exact in both forms), so its ABSOLUTE error against np.tanh in fp64 is measured directly, over 20 decades
    of |x| incl. the saturated ends (exp2 overflows to inf -> rcp 0 -> 1) and signed zeros: <= 3e-7 per evaluation (measured
    1.2e-7: the parity budget is 1e-5)."""
    from multiagent_gnn_policies_amd.learner.actor import Actor
    N, K, B = 128, 1, 8
    mags = np.concatenate([[0.0], np.logspace(-12, 2.2, B * N // 2 - 3), [88.0, 1e4, 1e38]])
    xs = np.concatenate([mags, -mags]).astype(np.float32)[:B * N].reshape(B, N)
    X = np.zeros((B, K, 6, N), dtype=np.float32)
    X[:, 0, 0] = xs
    X[:, 0, 1] = xs[:, ::-1]
    G = np.broadcast_to(np.eye(N, dtype=np.float32), (B, K, N, N)).copy()
    actor = Actor(6, 2, list(hidden), K, 0).cuda()
    with torch.no_grad():
        for conv in actor.conv_layers:
            conv.weight.zero_(); conv.bias.zero_()
        for conv in actor.conv_layers:                       # channel 0 -> 0, 1 -> 1 through every layer
            conv.weight[0, 0, 0, 0] = 1.0; conv.weight[1, 1, 0, 0] = 1.0
        out = actor(torch.from_numpy(X).cuda(), torch.from_numpy(G).cuda()).cpu().numpy().astype(np.float64)
    ref0, ref1 = xs.astype(np.float64), xs[:, ::-1].astype(np.float64)
    for _ in hidden:
        ref0, ref1 = np.tanh(ref0), np.tanh(ref1)
    err = max(np.max(np.abs(out[:, 0, 0] - ref0)), np.max(np.abs(out[:, 0, 1] - ref1)))
    print('tanh_fast through %d hidden layer(s): max abs error %.3g' % (len(hidden), err))
    assert np.all(np.isfinite(out)) and err <= 3e-7 * len(hidden)
