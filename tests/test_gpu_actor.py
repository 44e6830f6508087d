"""GPU parity of the drop-in classes (Actor, MultiAgentStateWithDelay, DAGGER pieces) against the golden
vectors the REFERENCE produced (tests/golden), and against the numpy oracle on fresh inputs."""
import configparser

import numpy as np
import pytest
import torch

from conftest import (ACTOR_GOLDENS, STATE_GOLDENS, DAGGER_GOLDENS, load_golden, golden_weights, golden_grads,
                      golden_inputs)
from oracle import actor as oa, synth

pytestmark = pytest.mark.gpu
TOL = 1e-5


def relerr(a, b):
    a = np.asarray(a, dtype=np.float64); b = np.asarray(b, dtype=np.float64)
    return float(np.max(np.abs(a - b) / np.maximum(1.0, np.abs(b))))


def build_actor(g):
    from multiagent_gnn_policies_amd.learner import Actor
    B, K, F, N = [int(v) for v in g['shape']]
    Ws, bs = golden_weights(g)
    hidden = [int(h) for h in g['hidden']]
    n_a = Ws[-1].shape[0]
    a = Actor(F, n_a, hidden, K, int(g['ind_agg']))
    sd = {}
    for i, (W, b) in enumerate(zip(Ws, bs)):
        sd[f'conv_layers.{i}.weight'] = torch.from_numpy(W)
        sd[f'conv_layers.{i}.bias'] = torch.from_numpy(b)
    a.load_state_dict(sd)
    return a.to('cuda')


@pytest.mark.parametrize('fused', [True, False])
@pytest.mark.parametrize('name', ACTOR_GOLDENS)
def test_actor_forward_backward_vs_reference(name, fused):
    g = load_golden(name)
    X, G = golden_inputs(g)
    actor = build_actor(g)
    actor.use_fused = fused
    xt = torch.from_numpy(X).cuda()
    gt = torch.from_numpy(G).cuda()
    out = actor(xt, gt)
    assert out.shape == g['out'].shape
    # Parity bar: 1e-5 (relative to max(1,|ref|)) against the reference.  The reference is itself an fp32
    # evaluation; where ITS distance to the exact (fp64) result is not negligible (the dense stress input drives
    # the checkpoint to |out| = 36 and the reference is 8.7e-6 from exact), that distance is added to the bound,
    # and the GPU result must additionally be as close to the exact result as the reference is (+ 1e-5).
    Ws, bs = golden_weights(g)
    exact = oa.forward(X, G, Ws, bs, int(g['ind_agg']), dtype=np.float64)
    ref_noise = relerr(g['out'], exact)
    got = out.detach().cpu().numpy()
    assert relerr(got, g['out']) <= TOL + ref_noise
    assert relerr(got, exact) <= TOL + ref_noise
    if not int(g['dense']):
        assert relerr(got, g['out']) <= TOL          # realistic operators: the plain 1e-5 bar holds
        # ... and it holds as an ABSOLUTE bound (max|gpu - cpu| <= 1e-5), elementwise, even where
        # the shipped checkpoint drives outputs to |33| (1 ulp = 3.8e-6 there); against the exact result the kernel is
        # within 4 ulp of the largest output
        abs_err = float(np.max(np.abs(got.astype(np.float64) - g['out'].astype(np.float64))))
        ulp = float(np.spacing(np.float32(np.max(np.abs(g['out'])))))
        print('%s fused=%s: max abs err vs reference %.3g (%.2f ulp of max|out| = %.3g), vs exact %.3g' % (
            name, fused, abs_err, abs_err / ulp, float(np.max(np.abs(g['out']))), float(np.max(np.abs(got - exact)))))
        assert abs_err <= 1e-5
        assert float(np.max(np.abs(got - exact))) <= max(4.0 * ulp, 2e-6)
    # parameter gradients of mse_loss against the reference's autograd
    from multiagent_gnn_policies_amd import ops
    loss = ops.mse_loss(out, torch.from_numpy(g['target']).cuda())
    assert abs(loss.item() - float(g['loss'])) <= 1e-5 * max(1.0, float(g['loss']))
    loss.backward()
    gWs, gbs = golden_grads(g)
    for i, conv in enumerate(actor.conv_layers):
        assert relerr(conv.weight.grad.cpu().numpy(), gWs[i]) <= 2e-5
        assert relerr(conv.bias.grad.cpu().numpy(), gbs[i]) <= 2e-5


@pytest.mark.parametrize('name', ACTOR_GOLDENS)
def test_actor_input_gradient_vs_reference(name):
    """dL/d delay_state through the composed path (aggregation backward)."""
    g = load_golden(name)
    X, G = golden_inputs(g)
    actor = build_actor(g)
    xt = torch.from_numpy(X).cuda().requires_grad_(True)
    out = actor(xt, torch.from_numpy(G).cuda())
    from multiagent_gnn_policies_amd import ops
    ops.mse_loss(out, torch.from_numpy(g['target']).cuda()).backward()
    assert relerr(xt.grad.cpu().numpy(), g['dX']) <= 2e-5


def test_actor_shape_asserts_match_reference():
    from multiagent_gnn_policies_amd.learner import Actor
    a = Actor(6, 2, [32, 32], 3, 0).cuda()
    X = torch.zeros(2, 3, 6, 10, device='cuda'); G = torch.zeros(2, 3, 10, 10, device='cuda')
    a(X, G)
    with pytest.raises(AssertionError):
        a(X, torch.zeros(1, 3, 10, 10, device='cuda'))
    with pytest.raises(AssertionError):
        a(torch.zeros(2, 2, 6, 10, device='cuda'), G)
    with pytest.raises(AssertionError):
        a(torch.zeros(2, 3, 5, 10, device='cuda'), G)
    with pytest.raises(AssertionError):
        a(X, torch.zeros(2, 3, 10, 9, device='cuda'))


def _args(**kw):
    cp = configparser.ConfigParser()
    base = dict(alg='dagger', batch_size='20', buffer_size='10000', updates_per_step='200', seed='11',
                actor_lr='5e-5', n_train_episodes='400', beta_coeff='0.993', test_interval='40',
                n_test_episodes='20', k='3', hidden_size='32', gamma='0.99', tau='0.5',
                env='FlockingRelative-v0', v_max='3.0', comm_radius='1.0', n_agents='100',
                n_actions='2', n_states='6', debug='False', dt='0.01')
    base.update({k: str(v) for k, v in kw.items()})
    cp['DEFAULT'] = base
    cp['test'] = {}
    return cp['test']


@pytest.mark.parametrize('name', STATE_GOLDENS)
def test_state_with_delay_vs_reference(name):
    from multiagent_gnn_policies_amd.learner import MultiAgentStateWithDelay
    g = load_golden(name)
    n, k, steps = int(g['n']), int(g['k']), int(g['steps'])
    args = _args(n_agents=n, k=k)
    rs = np.random.RandomState(1234 + n + k)
    prev = None
    for t in range(steps):
        vals = rs.randn(n, 6)
        net = synth.geometric_adjacency(rs, n)
        st = MultiAgentStateWithDelay(torch.device('cuda:0'), args, (vals, net), prev_state=prev)
        assert st.values.shape == (1, 1, 6, n) and st.network.shape == (1, 1, n, n)
        assert st.delay_gso.shape == (1, k, n, n) and st.delay_state.shape == (1, k, 6, n)
        if f'delay_gso_{t}' in g:
            assert relerr(st.delay_gso.cpu().numpy(), g[f'delay_gso_{t}']) <= 1e-6
            assert np.array_equal(st.delay_state.cpu().numpy(), g[f'delay_state_{t}'])
            assert relerr(st.curr_gso.cpu().numpy(), g[f'curr_gso_{t}']) <= 1e-6
        cs = synth.checksum(st.delay_gso.cpu().numpy(), st.delay_state.cpu().numpy(), st.curr_gso.cpu().numpy())
        assert abs(cs - float(g[f'cs_{t}'])) <= 1e-4 * max(1.0, abs(cs))
        prev = st


UPDATE_PATHS = {                                   # (use_graphed_update, use_train_step)
    'graph_two_launch': (True, True),              # mgp_train_step replayed from a HIP graph (the default)
    'graph_five_launch': (True, False),            # fwd / mse / bwd / scatter / adam graph
    'eager_train_grads': (False, True),            # mgp_train_grads + mgp_adam_step (the data-parallel form)
    'eager_composed': (False, False),              # autograd over the separate kernels
}


@pytest.mark.parametrize('path', sorted(UPDATE_PATHS))
@pytest.mark.parametrize('name', DAGGER_GOLDENS)
def test_dagger_learner_vs_reference(name, path):
    """select_action and three gradient_steps of the DAGGER learner against the reference's, on every update path."""
    from types import SimpleNamespace
    from multiagent_gnn_policies_amd.learner.gnn_dagger import DAGGER
    from multiagent_gnn_policies_amd.learner import Transition
    g = load_golden(name)
    n, k, bsz = int(g['n']), int(g['k']), int(g['bsz'])
    args = _args(n_agents=n, k=k, batch_size=bsz)
    torch.manual_seed(11)
    learner = DAGGER(torch.device('cuda:0'), args)
    learner.use_graphed_update, learner.use_train_step = UPDATE_PATHS[path]
    Ws, bs = golden_weights(g, 'w0__')
    for i, conv in enumerate(learner.actor.conv_layers):      # same seed => identical default init
        assert np.array_equal(conv.weight.detach().cpu().numpy(), Ws[i])
        assert np.array_equal(conv.bias.detach().cpu().numpy(), bs[i])
    X1, G1 = synth.make_inputs(77, 1, k, 6, n)
    st = SimpleNamespace(delay_state=torch.from_numpy(X1).cuda(), delay_gso=torch.from_numpy(G1).cuda())
    act = learner.select_action(st)
    assert act.shape == (n, 2)
    assert relerr(act.cpu().numpy(), g['select_action']) <= TOL
    for step in range(3):
        X, G = synth.make_inputs(200 + step, bsz, k, 6, n)
        labels = np.random.RandomState(300 + step).randn(bsz, 1, 2, n).astype(np.float32)
        states = [SimpleNamespace(delay_state=torch.from_numpy(X[i:i + 1]).cuda(),
                                  delay_gso=torch.from_numpy(G[i:i + 1]).cuda()) for i in range(bsz)]
        actions = [torch.from_numpy(labels[i:i + 1]).cuda() for i in range(bsz)]
        loss = learner.gradient_step(Transition(tuple(states), tuple(actions), None, None, None))
        assert abs(loss - g['losses'][step]) <= 1e-5
        rWs, rbs = golden_weights(g, f'w{step + 1}__')
        for i, conv in enumerate(learner.actor.conv_layers):
            assert np.max(np.abs(conv.weight.detach().cpu().numpy() - rWs[i])) <= 2e-6
            assert np.max(np.abs(conv.bias.detach().cpu().numpy() - rbs[i])) <= 2e-6
