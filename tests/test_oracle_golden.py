"""CPU: the numpy oracle (oracle/) against golden vectors produced by the reference itself
(tests/golden/gen_golden.py).  This is what pins the oracle."""
import numpy as np
import pytest

from conftest import (ACTOR_GOLDENS, STATE_GOLDENS, DAGGER_GOLDENS, load_golden, golden_weights,
                      golden_grads, golden_inputs)
from oracle import actor as oa, state as os_, dagger as od, synth

# fp32 tolerance: the north star asks 1e-5 on fp32; outputs with the shipped checkpoint reach |33|
# (1 ulp = 3.8e-6), so the bound is applied relative to max(1, |ref|).
ATOL = 1e-5


def close(a, b, tol=ATOL):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return np.max(np.abs(a - b) / np.maximum(1.0, np.abs(b))) <= tol


@pytest.mark.parametrize('name', ACTOR_GOLDENS)
def test_actor_forward_matches_reference(name):
    g = load_golden(name)
    X, G = golden_inputs(g)
    Ws, bs = golden_weights(g)
    out32 = oa.forward(X, G, Ws, bs, int(g['ind_agg']), dtype=np.float32)
    out64 = oa.forward(X, G, Ws, bs, int(g['ind_agg']), dtype=np.float64)
    assert out32.shape == g['out'].shape
    assert close(out32, g['out'])
    assert close(out64, g['out'])


@pytest.mark.parametrize('name', ACTOR_GOLDENS)
def test_actor_backward_matches_reference(name):
    g = load_golden(name)
    X, G = golden_inputs(g)
    Ws, bs = golden_weights(g)
    gWs, gbs = golden_grads(g)
    ia = int(g['ind_agg'])
    out, cache = oa.forward(X, G, Ws, bs, ia, dtype=np.float64, return_cache=True)
    assert abs(od.mse_loss(out, g['target']) - float(g['loss'])) <= 1e-5 * max(1.0, float(g['loss']))
    d_out = od.mse_grad(out, g['target'])
    dWs, dbs, dX = oa.backward(d_out, G, Ws, ia, cache, need_dx=True)
    for a, b in zip(dWs + dbs, gWs + gbs):
        assert a.shape == b.shape
        scale = max(1.0, float(np.abs(b).max()))
        assert np.max(np.abs(a - b)) <= 2e-5 * scale
    assert np.max(np.abs(dX - g['dX'])) <= 2e-5 * max(1.0, float(np.abs(g['dX']).max()))


@pytest.mark.parametrize('name', STATE_GOLDENS)
def test_state_recursion_matches_reference(name):
    g = load_golden(name)
    n, k, steps = int(g['n']), int(g['k']), int(g['steps'])
    full = 'network_0' in g
    rs = np.random.RandomState(1234 + n + k)
    Gp = Xp = None
    for t in range(steps):
        vals = rs.randn(n, 6)
        net = synth.geometric_adjacency(rs, n)
        assert np.array_equal(vals, g[f'values_{t}'])
        if full:
            assert np.array_equal(net, g[f'network_{t}'])
        v, a = os_.cast_env_state(vals, net)
        Gn, Xn = os_.gso_update(a[0], Gp, v[0], Xp, k)
        C = os_.gso_powers(a[0], k)
        if f'delay_gso_{t}' in g:
            assert np.max(np.abs(Gn - g[f'delay_gso_{t}'])) <= 1e-6
            assert np.array_equal(Xn, g[f'delay_state_{t}'])
            assert np.max(np.abs(C - g[f'curr_gso_{t}'])) <= 1e-6
            if t > 0 and k > 1:
                # delay_gso[1] == A_t exactly (A @ I), reference state_with_delay.py:47
                assert np.array_equal(g[f'delay_gso_{t}'][0, 1], a[0, 0])
        cs = synth.checksum(Gn, Xn, C)
        assert abs(cs - float(g[f'cs_{t}'])) <= 1e-4 * max(1.0, abs(cs))
        Gp, Xp = Gn, Xn


@pytest.mark.parametrize('name', DAGGER_GOLDENS)
def test_dagger_update_matches_reference(name):
    g = load_golden(name)
    n, k, bsz, lr = int(g['n']), int(g['k']), int(g['bsz']), float(g['lr'])
    Ws, bs = golden_weights(g, 'w0__')
    # select_action
    X1, G1 = synth.make_inputs(77, 1, k, 6, n)
    assert abs(synth.checksum(X1, G1) - float(g['select_cs'])) < 1e-6
    act = od.action_from_output(oa.forward(X1, G1, Ws, bs, 0))
    assert act.shape == (n, 2)
    assert np.max(np.abs(act - g['select_action'])) <= ATOL
    # three Adam steps
    m = [np.zeros_like(p) for pair in zip(Ws, bs) for p in pair]
    v = [np.zeros_like(p) for p in m]
    for step in range(3):
        X, G = synth.make_inputs(200 + step, bsz, k, 6, n)
        labels = np.random.RandomState(300 + step).randn(bsz, 1, 2, n).astype(np.float32)
        loss, Ws, bs, m, v, _ = od.gradient_step(X, G, labels, Ws, bs, 0, m, v, step + 1, lr)
        assert abs(loss - g['losses'][step]) <= 1e-5
        rWs, rbs = golden_weights(g, f'w{step + 1}__')
        for a, b in zip(Ws + bs, rWs + rbs):
            # Adam's first steps move each weight by ~lr regardless of gradient scale; a sign-level
            # disagreement on a ~0 gradient would show as 2*lr. Require far better than that.
            assert np.max(np.abs(a - b)) <= 2e-6


def test_label_action_roundtrip():
    a = np.random.RandomState(0).randn(7, 2)
    lab = od.label_from_action(a)
    assert lab.shape == (1, 1, 2, 7)
    assert np.array_equal(od.action_from_output(lab), a)


@pytest.mark.parametrize('name', [n for n in ACTOR_GOLDENS])
def test_torch_port_matches_reference(name):
    """The torch-CPU port timed as bench.py's cpu_baseline reproduces the reference outputs."""
    import torch
    from oracle import torch_port
    g = load_golden(name)
    X, G = golden_inputs(g)
    Ws, bs = golden_weights(g)
    out = torch_port.actor_forward(torch.from_numpy(X), torch.from_numpy(G), [torch.from_numpy(w) for w in Ws],
                                   [torch.from_numpy(b) for b in bs], int(g['ind_agg']), int(g['shape'][1]))
    assert close(out.numpy(), g['out'], 2e-6)


def test_torch_port_state_matches_reference():
    from oracle import torch_port
    g = load_golden('state_N16_K3')
    prev = None
    for t in range(int(g['steps'])):
        st = torch_port.PortState(g[f'values_{t}'], g[f'network_{t}'], 3, prev)
        assert np.array_equal(st.delay_gso.numpy(), g[f'delay_gso_{t}'])
        assert np.array_equal(st.delay_state.numpy(), g[f'delay_state_{t}'])
        assert np.array_equal(st.curr_gso.numpy(), g[f'curr_gso_{t}'])
        prev = st
