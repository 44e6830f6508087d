"""GPU: randomised shapes.  Actor forward/backward (fused where covered, composed otherwise) and the state / sim kernels
against the fp64 oracle on 120 / 40 random configurations -- odd N, N not a multiple of 4, F != 6, 0-4 hidden layers,
every ind_agg, K 1-4."""
import numpy as np
import pytest
import torch

from oracle import actor as oa, state as os_, dagger as od, flock as ofl, synth

from conftest import reference_noise, check_parity

pytestmark = pytest.mark.gpu


def relerr(a, b):
    a = np.asarray(a, dtype=np.float64); b = np.asarray(b, dtype=np.float64)
    return float(np.max(np.abs(a - b) / np.maximum(1.0, np.abs(b)))) if a.size else 0.0


def random_actor_case(rs):
    B = int(rs.randint(1, 5)); K = int(rs.randint(1, 5)); F = int(rs.choice([1, 2, 3, 6, 6, 6, 8, 11]))
    N = int(rs.choice([1, 3, 7, 16, 31, 64, 100, 100, 129, 180, 257]))
    n_hidden = int(rs.randint(0, 5))
    hidden = [int(rs.choice([1, 4, 5, 16, 32, 32, 48, 64, 96, 128])) for _ in range(n_hidden)]
    n_a = int(rs.choice([1, 2, 2, 3]))
    ind_agg = int(rs.randint(0, n_hidden + 1))
    return B, K, F, N, hidden, n_a, ind_agg


@pytest.mark.parametrize('seed', range(120))
def test_actor_random_configuration(seed):
    from multiagent_gnn_policies_amd.learner import Actor
    from multiagent_gnn_policies_amd import ops
    rs = np.random.RandomState(1000 + seed)
    B, K, F, N, hidden, n_a, ind_agg = random_actor_case(rs)
    torch.manual_seed(seed)
    actor = Actor(F, n_a, hidden, K, ind_agg).cuda()
    X, G = (synth.make_dense_inputs if seed % 2 else synth.make_inputs)(seed, B, K, F, N)
    Ws = [c.weight.detach().cpu().numpy() for c in actor.conv_layers]
    bs = [c.bias.detach().cpu().numpy() for c in actor.conv_layers]
    ref, cache = oa.forward(X, G, Ws, bs, ind_agg, dtype=np.float64, return_cache=True)
    target = rs.randn(*ref.shape).astype(np.float32)
    d_out = od.mse_grad(ref, target)
    dWs, dbs, dX = oa.backward(d_out, G, Ws, ind_agg, cache, need_dx=True)
    for fused in (True, False):
        actor.use_fused = fused
        actor.zero_grad()
        xt = torch.from_numpy(X).cuda().requires_grad_(not fused)        # input grads force the composed path
        out = actor(xt, torch.from_numpy(G).cuda())
        assert out.shape == ref.shape
        assert relerr(out.detach().cpu().numpy(), ref) <= 1e-5, (B, K, F, N, hidden, n_a, ind_agg, fused)
        ops.mse_loss(out, torch.from_numpy(target).cuda()).backward()
        for i, conv in enumerate(actor.conv_layers):
            assert relerr(conv.weight.grad.cpu().numpy(), dWs[i]) <= 2e-5
            assert relerr(conv.bias.grad.cpu().numpy(), dbs[i]) <= 2e-5
        if not fused:
            assert relerr(xt.grad.cpu().numpy(), dX) <= 2e-5


MFMA_SHAPES = [(N, K) for N in (16, 20, 36, 60, 64, 68, 96, 100, 104, 124, 128) for K in (1, 2, 3, 4, 6)
               if K * (2 if N > 64 else 1) <= 6]


@pytest.mark.parametrize('N,K', MFMA_SHAPES)
def test_actor_fwd_mfma_variant_shapes(N, K):
    """Every (N, K) class the MFMA-aggregation variant of mgp_actor_fwd takes (N % 4 == 0, 16 <= N <= 128, K * column blocks
    <= 6: one and two column blocks, partial last blocks, row steps past N, every tap count), F and widths varied with
    the shape, forward AND the activations it saves for backward, against the fp64 oracle."""
    from multiagent_gnn_policies_amd.learner import Actor
    from multiagent_gnn_policies_amd import ops
    seed = 31 * N + K
    rs = np.random.RandomState(seed)
    F = int(rs.choice([1, 3, 6, 8])); B = int(rs.choice([1, 3, 9]))
    hidden = [int(rs.choice([4, 16, 20, 32, 64, 80, 128])) for _ in range(int(rs.randint(0, 4)))]
    torch.manual_seed(seed)
    actor = Actor(F, 2, hidden, K, 0).cuda()
    actor.use_fused = True
    X, G = (synth.make_dense_inputs if seed % 2 else synth.make_inputs)(seed, B, K, F, N)
    Ws = [c.weight.detach().cpu().numpy() for c in actor.conv_layers]
    bs = [c.bias.detach().cpu().numpy() for c in actor.conv_layers]
    ref, cache = oa.forward(X, G, Ws, bs, 0, dtype=np.float64, return_cache=True)
    target = rs.randn(*ref.shape).astype(np.float32)
    dWs, dbs, _ = oa.backward(od.mse_grad(ref, target), G, Ws, 0, cache, need_dx=False)
    out = actor(torch.from_numpy(X).cuda(), torch.from_numpy(G).cuda())
    assert relerr(out.detach().cpu().numpy(), ref) <= 1e-5, (B, K, F, N, hidden)
    ops.mse_loss(out, torch.from_numpy(target).cuda()).backward()
    for i, conv in enumerate(actor.conv_layers):
        assert relerr(conv.weight.grad.cpu().numpy(), dWs[i]) <= 2e-5
        assert relerr(conv.bias.grad.cpu().numpy(), dbs[i]) <= 2e-5
    # LDS left over by other kernels must not leak in through padding (NaN-poisoned run-to-run determinism check)
    out2 = actor(torch.from_numpy(X).cuda(), torch.from_numpy(G).cuda())
    assert torch.equal(out, out2)


WIDE_SHAPES = [((128, 128), 100, 3), ((128, 128), 128, 3), ((128, 128), 16, 5), ((128, 128), 64, 1), ((64, 64), 100, 3),
               ((64, 64), 128, 2), ((48, 100), 100, 3), ((128, 36), 68, 2), ((36, 32), 100, 3), ((32, 40), 20, 4),
               ((100, 72), 124, 3), ((64, 128), 96, 1), ((96, 64), 36, 5)]


@pytest.mark.parametrize('hidden,N,K', WIDE_SHAPES)
def test_actor_fwd_wide_inference_kernel(hidden, N, K):
    """actor_fwd_wide_kernel: inference through mgp_actor_fwd (no saved activations) with TWO hidden layers of which one is
    wider than 32 -- split-bf16 layers at 64 or 128 padded channels, the last planes of the [128, 128] weight image stored
    behind the first barrier -- against the fp64 oracle and against the generic fp32-MFMA chain (MGP_ACTOR_WIDE=0 is read once
    per process, so the generic form is reached through the training forward, which saves activations); widths that are not
    multiples of 16, one and two column blocks, K 1..5, repeated launches bit-identical (padding never read)."""
    from multiagent_gnn_policies_amd.learner import Actor
    seed = 7 * N + K + hidden[0]
    rs = np.random.RandomState(seed)
    B = int(rs.choice([1, 3, 9]))
    torch.manual_seed(seed)
    actor = Actor(6, 2, list(hidden), K, 0).cuda()
    actor.use_fused = True
    with torch.no_grad():                                      # weights and biases large enough to saturate some tanh units
        for c in actor.conv_layers:
            c.weight.mul_(2.0); c.bias.add_(0.1 * torch.randn_like(c.bias))
    X, G = (synth.make_dense_inputs if seed % 2 else synth.make_inputs)(seed, B, K, 6, N)
    Ws = [c.weight.detach().cpu().numpy() for c in actor.conv_layers]
    bs = [c.bias.detach().cpu().numpy() for c in actor.conv_layers]
    ref = oa.forward(X, G, Ws, bs, 0, dtype=np.float64)
    xt, gt = torch.from_numpy(X).cuda(), torch.from_numpy(G).cuda()
    with torch.no_grad():
        out = actor(xt, gt)
        out2 = actor(xt, gt)
    generic = actor(xt, gt)                                    # grad mode: the generic chain (it saves activations)
    assert out.shape == ref.shape and torch.equal(out, out2)
    e_wide, e_gen = relerr(out.cpu().numpy(), ref), relerr(generic.detach().cpu().numpy(), ref)
    print('hidden %s N %d K %d B %d: wide kernel %.2e, generic chain %.2e vs fp64' % (hidden, N, K, B, e_wide, e_gen))
    assert e_wide <= 1e-5 and e_gen <= 1e-5, (hidden, N, K, B)


DEEP_SHAPES = [((128, 128, 128), 100, 3), ((128, 128, 128, 128), 100, 3), ((128, 128, 128), 128, 3), ((128, 64, 128), 16, 5),
               ((100, 72, 96, 48), 124, 2), ((32, 128, 32), 64, 1), ((128, 128, 128, 128, 128), 36, 4), ((68, 128, 128), 96, 3)]


@pytest.mark.parametrize('hidden,N,K', DEEP_SHAPES)
def test_actor_fwd_deep_inference(hidden, N, K):
    """mgp_actor_fwd_deep (three or more hidden layers, one wider than 64: cfg/hidden_size.cfg n_layers 3, 4 at hidden_size
    128): actor_fwd_wide_kernel with every further hidden layer in the same launch (the weight image in LDS rebuilt per layer,
    activations in the accumulators), against the fp64 oracle and against the composed path (mgp_agg_fwd + mgp_dense_fwd per
    layer, what these shapes ran on before); ragged widths, one and two column blocks, up to five hidden layers, repeated calls
    bit-identical."""
    from multiagent_gnn_policies_amd.learner import Actor
    seed = 5 * N + K + len(hidden)
    rs = np.random.RandomState(seed)
    B = int(rs.choice([1, 3, 9]))
    torch.manual_seed(seed)
    actor = Actor(6, 2, list(hidden), K, 0).cuda()
    with torch.no_grad():
        for c in actor.conv_layers:
            c.weight.mul_(1.5); c.bias.add_(0.1 * torch.randn_like(c.bias))
    X, G = (synth.make_dense_inputs if seed % 2 else synth.make_inputs)(seed, B, K, 6, N)
    Ws = [c.weight.detach().cpu().numpy() for c in actor.conv_layers]
    bs = [c.bias.detach().cpu().numpy() for c in actor.conv_layers]
    ref = oa.forward(X, G, Ws, bs, 0, dtype=np.float64)
    xt, gt = torch.from_numpy(X).cuda(), torch.from_numpy(G).cuda()
    from multiagent_gnn_policies_amd.learner import actor_fused
    with torch.no_grad():
        actor.use_fused = True
        import ctypes
        cdims = (ctypes.c_int * (len(hidden) + 2))(6, *hidden, 2)
        deep = actor_fused._try_forward_deep(actor, xt, gt, cdims)      # (shapes the one-launch plan also covers go there in actor())
        assert deep is not None, 'shape not taken by mgp_actor_fwd_deep'
        deep2 = actor_fused._try_forward_deep(actor, xt, gt, cdims)
        out = actor(xt, gt)
        out2 = actor(xt, gt)
        actor.use_fused = False
        composed = actor(xt, gt)
    assert out.shape == ref.shape and torch.equal(out, out2) and torch.equal(deep, deep2)
    assert relerr(out.cpu().numpy(), ref) <= 1e-5
    e_deep, e_comp = relerr(deep.cpu().numpy(), ref), relerr(composed.cpu().numpy(), ref)
    print('hidden %s N %d K %d B %d: deep path %.2e, composed %.2e vs fp64' % (hidden, N, K, B, e_deep, e_comp))
    assert e_deep <= 1e-5 and e_comp <= 1e-5, (hidden, N, K, B)


@pytest.mark.parametrize('seed', range(40))
def test_state_and_sim_random_sizes(seed):
    from multiagent_gnn_policies_amd.envs import FlockParams, VecFlock
    from multiagent_gnn_policies_amd.learner.state_with_delay import BatchedDelayState
    rs = np.random.RandomState(2000 + seed)
    B = int(rs.randint(1, 4)); K = int(rs.randint(1, 5)); N = int(rs.choice([5, 12, 17, 36, 50, 100, 128, 130, 200]))
    op = ofl.FlockParams(n_agents=N, init_mode='grid', comm_radius=float(rs.choice([0.8, 1.0, 1.5])),
                         mean_pooling=bool(rs.randint(0, 2)), n_leaders=int(rs.randint(0, 3)),
                         centralized=bool(rs.randint(0, 2)))
    p = FlockParams(**{f: getattr(op, f) for f in FlockParams.__dataclass_fields__})
    xs = np.stack([ofl.sample_candidate_grid(rs, op) for _ in range(B)])
    sim = VecFlock(B, p, 'cuda', with_expert=True)
    sim.set_state(xs)
    st = BatchedDelayState('cuda', B, K, 6, N)
    st.push(sim.network, sim.features)
    h = [ofl.helpers(xs[b], op) for b in range(B)]
    Gp, Xp = os_.gso_update(np.stack([q['network'] for q in h]).astype(np.float32), None,
                            np.stack([q['values'].T for q in h]).astype(np.float32), None, K, dtype=np.float64)
    for t in range(K + 1):
        u = rs.uniform(-1.2, 1.2, size=(B, N, 2)).astype(np.float32)
        sim.step_advance(torch.from_numpy(u).cuda(), st)
        xs = np.stack([ofl.integrate(xs[b], u[b], op) for b in range(B)])
        assert np.array_equal(sim.x.cpu().numpy(), xs)
        h = [ofl.helpers(xs[b], op) for b in range(B)]
        A = np.stack([q['network'] for q in h]).astype(np.float32)
        Xt = np.stack([q['values'].T for q in h]).astype(np.float32)
        Gp, Xp = os_.gso_update(A.astype(np.float64), Gp, Xt, Xp, K, dtype=np.float64)
        assert relerr(st.delay_gso.cpu().numpy(), Gp) <= 1e-5
        assert relerr(st.delay_state.cpu().numpy(), Xp) <= 1e-6
        for b in range(B):
            assert relerr(sim.expert[b].cpu().numpy(), ofl.controller(xs[b], op)) <= 1e-6
            assert abs(sim.reward[b].item() - ofl.reward(xs[b], op)) <= 1e-12 * max(1.0, abs(ofl.reward(xs[b], op)))


@pytest.mark.parametrize('seed', range(60))
def test_train_grads_random_configuration(seed):
    """mgp_train_grads (forward + MSE + parameter backward in one tile kernel) on random shapes vs the oracle."""
    import ctypes
    from multiagent_gnn_policies_amd import _lib, ops
    from multiagent_gnn_policies_amd.learner.actor_fused import _ptr_array
    rs = np.random.RandomState(9000 + seed)
    B = int(rs.choice([1, 2, 5, 20, 33])); K = int(rs.randint(1, 5)); F = int(rs.choice([2, 6, 6, 8]))
    N = int(rs.choice([3, 16, 17, 50, 100, 100, 129, 200]))
    hidden = [int(rs.choice([1, 4, 16, 32, 32, 48, 64, 96, 128])) for _ in range(int(rs.randint(0, 4)))]
    if sum(h > 64 for h in hidden) > 2:
        hidden = hidden[:2]                                      # three layers wider than 64: beyond the LDS
    n_a = int(rs.choice([1, 2, 2, 3]))
    dims = [F] + hidden + [n_a]
    L = _lib.lib()
    cd = (ctypes.c_int * len(dims))(*dims)
    if not L.mgp_train_supported(cd, len(dims) - 1, B, K, N):
        assert max(hidden) > 64                                     # only layers wider than 64 can outgrow the LDS plan
        pytest.skip('LDS plan of this wide configuration does not fit')
    X, G = (synth.make_dense_inputs if seed % 2 else synth.make_inputs)(seed, B, K, F, N)
    Ws, bs = [], []
    for i in range(len(dims) - 1):
        cin = dims[i] * (K if i == 0 else 1)
        Ws.append((rs.randn(dims[i + 1], dims[i], K if i == 0 else 1, 1) / np.sqrt(cin)).astype(np.float32))
        bs.append((0.1 * rs.randn(dims[i + 1])).astype(np.float32))
    ref, cache = oa.forward(X, G, Ws, bs, 0, dtype=np.float64, return_cache=True)
    target = rs.randn(*ref.shape).astype(np.float32)
    dWs, dbs, _ = oa.backward(od.mse_grad(ref, target), G, Ws, 0, cache, need_dx=False)
    flat_ref = np.concatenate([np.concatenate([dWs[i].ravel(), dbs[i].ravel()]) for i in range(len(Ws))])
    Wd = [torch.from_numpy(w).cuda() for w in Ws]; bd = [torch.from_numpy(b).cuda() for b in bs]
    flat = torch.full((flat_ref.size,), float('nan'), device='cuda')
    loss = torch.zeros((1,), device='cuda')
    ws = torch.zeros((L.mgp_train_workspace(cd, len(Ws), B, K, N),), device='cuda')
    Xd, Gd, Td = torch.from_numpy(X).cuda(), torch.from_numpy(G).cuda(), torch.from_numpy(target).cuda()
    for rep in range(2):                               # second call: same workspace, bit-identical result
        _lib.check(L.mgp_train_grads(ops._ptr(Xd), ops._ptr(Gd), ops._ptr(Td), _ptr_array(Wd), _ptr_array(bd), cd,
                                     len(Ws), ops._ptr(flat), ops._ptr(loss), ops._ptr(ws), B, K, N, ops._stream()),
                   'mgp_train_grads')
        got = flat.cpu().numpy()
        if rep:
            assert np.array_equal(got, first)
        first = got.copy()
    assert relerr(got, flat_ref) <= 2e-5, (B, K, F, N, hidden, n_a)
    assert abs(loss.item() - od.mse_loss(ref, target)) <= 1e-5 * max(1.0, od.mse_loss(ref, target))
    # the same update on the aggregated first-layer input (mgp_train_grads_agg): Z[b, f K + k] = X[b, k, f] . G[b, k]
    assert L.mgp_train_agg_supported(cd, len(Ws), B, K, N)
    Z = np.einsum('bkfm,bkmn->bfkn', X.astype(np.float64), G.astype(np.float64)).reshape(B, F * K, N).astype(np.float32)
    flat_a = torch.full((flat_ref.size,), float('nan'), device='cuda')
    loss_a = torch.zeros((1,), device='cuda')
    _lib.check(L.mgp_train_grads_agg(ops._ptr(torch.from_numpy(Z).cuda()), ops._ptr(Td), _ptr_array(Wd), _ptr_array(bd), cd, len(Ws),
                                     ops._ptr(flat_a), ops._ptr(loss_a), ops._ptr(ws), B, K, N, ops._stream()), 'mgp_train_grads_agg')
    assert relerr(flat_a.cpu().numpy(), flat_ref) <= 2e-5, (B, K, F, N, hidden, n_a)
    assert abs(loss_a.item() - od.mse_loss(ref, target)) <= 1e-5 * max(1.0, od.mse_loss(ref, target))


@pytest.mark.parametrize('K,hidden', [(1, (32, 32)), (2, (32, 32)), (3, (32, 32)), (4, (32, 32)), (3, (64, 64)), (3, (32, 32, 32)), (3, (16, 16))])
@pytest.mark.parametrize('B,N', [(20, 100), (3, 37), (20, 1000)])
def test_train_grads_agg_compiled_policy_shapes(K, hidden, B, N):
    """mgp_train_grads_agg on the shapes whose tile kernel has the policy compiled in (two hidden layers of 32 at K = 1..4,
    of 64 at K = 3: train_tile_kernel<true, CW, CFK>) and two generic neighbours, ragged N included, against fp64 autograd of
    mse_loss(MLP(Z), target) (reference gnn_dagger.py:85-93 behind actor.py:64-75's aggregation); a second call on the
    same workspace is bit-identical."""
    import ctypes
    from multiagent_gnn_policies_amd import _lib, ops
    from multiagent_gnn_policies_amd.learner.actor_fused import _ptr_array
    g = torch.Generator().manual_seed(100 * K + N + len(hidden))
    dims = (6,) + tuple(hidden) + (2,)
    nl = len(dims) - 1
    L = _lib.lib()
    cd = (ctypes.c_int * len(dims))(*dims)
    assert L.mgp_train_agg_supported(cd, nl, B, K, N)
    Z = torch.randn((B, 6 * K, N), generator=g)
    T = torch.randn((B, 2, N), generator=g)
    Ws = [torch.randn((dims[l + 1], 6 * K if l == 0 else dims[l]), generator=g) / np.sqrt(6 * K if l == 0 else dims[l]) for l in range(nl)]
    bs = [0.1 * torch.randn((dims[l + 1],), generator=g) for l in range(nl)]
    Wd = [w.double().requires_grad_(True) for w in Ws]; bd = [b_.double().requires_grad_(True) for b_ in bs]
    h = Z.double()
    for l in range(nl):
        h = torch.einsum('oc,scn->son', Wd[l], h) + bd[l][None, :, None]
        if l < nl - 1:
            h = torch.tanh(h)
    loss_ref = torch.nn.functional.mse_loss(h, T.double())
    loss_ref.backward()
    flat_ref = torch.cat([torch.cat([w.grad.reshape(-1), b_.grad.reshape(-1)]) for w, b_ in zip(Wd, bd)]).numpy()
    Wc = [w.cuda().contiguous() for w in Ws]; bc = [b_.cuda() for b_ in bs]
    ws = torch.zeros((L.mgp_train_workspace(cd, nl, B, K, N),), device='cuda')
    Zc, Tc = Z.cuda(), T.cuda()
    outs = []
    for rep in range(2):
        flat = torch.full((flat_ref.size,), float('nan'), device='cuda'); loss = torch.zeros((1,), device='cuda')
        _lib.check(L.mgp_train_grads_agg(ops._ptr(Zc), ops._ptr(Tc), _ptr_array(Wc), _ptr_array(bc), cd, nl, ops._ptr(flat),
                                         ops._ptr(loss), ops._ptr(ws), B, K, N, ops._stream()), 'mgp_train_grads_agg')
        outs.append((flat.cpu().numpy(), float(loss)))
    assert np.array_equal(outs[0][0], outs[1][0]) and outs[0][1] == outs[1][1]
    assert relerr(outs[0][0], flat_ref) <= 2e-5
    assert abs(outs[0][1] - float(loss_ref.detach())) <= 1e-5 * max(1.0, float(loss_ref.detach()))


@pytest.mark.parametrize('seed', range(40))
def test_resident_rollout_random_shapes(seed):
    """mgp_rollout_steps on random (N, K, layers, widths, spec variants): every step against the oracle transition."""
    from test_gpu_rollout import _make, _snapshot, _weights_np
    from multiagent_gnn_policies_amd import ops
    from multiagent_gnn_policies_amd.learner.rollouts import policy_rollout
    rs = np.random.RandomState(5000 + seed)
    N = int(rs.randint(8, 129)) if seed % 4 else int(rs.randint(129, 257))   # any N; N > 128 runs rollout_big_kernel
    K = int(rs.randint(1, 5))
    hidden = [(), (4,), (32,), (16, 16), (32, 32), (8, 32, 16), (32, 32, 32), (20, 12), (64, 64), (40,), (64, 8, 48)][int(rs.randint(0, 11))]
    variant = dict(mean_pooling=bool(rs.randint(0, 2)), n_leaders=int(rs.randint(0, 3)),
                   comm_radius=float(rs.choice([0.8, 1.0, 1.5])))
    B = int(rs.randint(1, 4))
    _, op, actor, sim, st = _make(N, K, hidden, B, seed=seed, **variant)
    action = torch.zeros((B, 1, 2, N), device='cuda')
    rewards = torch.zeros((B, 1), device='cuda', dtype=torch.float64)
    if not ops.rollout_supported(tuple(actor.layers), K, N):
        assert policy_rollout(actor, sim, st, 1, rewards=rewards, action=action) is False
        assert torch.isfinite(action).all()
        return
    Ws, bs = _weights_np(actor)
    for step in range(K + 1):
        x0, G0, X0 = _snapshot(sim, st)
        assert policy_rollout(actor, sim, st, 1, rewards=rewards, action=action)
        x1, G1, X1 = _snapshot(sim, st)
        u = action.cpu().numpy()
        # sum pooling at K = 4 makes operator entries O(100) and pre-activations O(1e4): there the fp32 evaluation of the
        # REFERENCE op sequence is itself further than 1e-5 from the exact result: conftest.NOISE_FACTOR x its own distance on top
        noise, ref = reference_noise(X0, G0, Ws, bs, K, per_episode=True)
        check_parity(u, ref, noise, 'fuzz N=%d K=%d hidden=%s step %d' % (N, K, hidden, step))
        for b in range(B):
            x_ref, vals, net, r = ofl.step(x0[b], u[b, 0].T.astype(np.float32), op)
            assert np.array_equal(x1[b], x_ref)
            if K > 1:
                assert np.array_equal(G1[b, 1], net.astype(np.float32))
            assert relerr(X1[b, 0], vals.T.astype(np.float32)) <= 1e-6
            Gr, Xr = os_.gso_update(net[None], G0[b:b + 1], vals.T[None].astype(np.float32), X0[b:b + 1], K, dtype=np.float64)
            assert relerr(G1[b], Gr[0]) <= 1e-6
            assert np.array_equal(X1[b, 1:], X0[b, :-1])
            assert abs(rewards[b, 0].item() - r) <= 1e-12 * max(1.0, abs(r))
