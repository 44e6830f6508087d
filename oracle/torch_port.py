"""PyTorch-CPU restatement of the reference's per-step op sequence (TEST ORACLE / CPU BASELINE ONLY).

This is what bench.py times as `cpu_baseline` ("kind": "port"): the same ATen ops, in the same order, that
the reference dispatches per environment step --
  state update   torch.zeros / eye / matmul / slice-assign      state_with_delay.py:34-53
  actor forward  permute / matmul / conv2d / tanh / view        actor.py:63-82
  action         permute + view                                 gnn_dagger.py:66-68
-- with the numpy fp64 simulator of oracle/flock.py standing in for gym_flock's env.step.
It is checked against the goldens in tests/test_oracle_golden.py::test_torch_port_matches_reference.
"""
import numpy as np
import torch
import torch.nn.functional as F

from . import flock as _flock


class PortState(object):
    """state_with_delay.py:6-53 (curr_gso included: the reference computes it every step)."""

    def __init__(self, values, network, k, prev=None, with_curr_gso=True):
        n, f = values.shape
        v = values.transpose(1, 0).reshape((1, 1, f, n))
        a = network.reshape((1, 1, n, n))
        self.values = torch.Tensor(v)
        self.network = torch.Tensor(a)
        if with_curr_gso:
            self.curr_gso = torch.zeros((1, k, n, n))
            self.curr_gso[0, 0] = torch.eye(n)
            for j in range(1, k):
                self.curr_gso[0, j] = torch.matmul(self.network, self.curr_gso[0, j - 1])
        self.delay_gso = torch.zeros((1, k, n, n))
        self.delay_gso[0, 0] = torch.eye(n)
        if prev is not None and k > 1:
            self.delay_gso[0, 1:k] = torch.matmul(self.network, prev.delay_gso[0, 0:k - 1])
        self.delay_state = torch.zeros((1, k, f, n))
        self.delay_state[0, 0] = self.values
        if prev is not None and k > 1:
            self.delay_state[0, 1:k] = prev.delay_state[0, 0:k - 1]


def actor_forward(delay_state, delay_gso, weights, biases, ind_agg, k):
    """actor.py:63-82 with F.conv2d standing in for the nn.Conv2d modules."""
    B, n = delay_state.shape[0], delay_state.shape[3]
    x = delay_state.permute(0, 2, 1, 3)
    n_layers = len(weights)
    for i in range(n_layers):
        if i == ind_agg:
            x = x.permute(0, 2, 1, 3)
            x = torch.matmul(x, delay_gso)
            x = x.permute(0, 2, 1, 3)
        step = k if i == ind_agg else 1
        x = F.conv2d(x, weights[i], biases[i], stride=(step, 1))
        if i < n_layers - 1:
            x = torch.tanh(x)
    return x.view((B, 1, weights[-1].shape[0], n))


def rollout_steps(x0, params, weights, biases, k, n_steps, with_curr_gso=True):
    """Reference-style single-episode loop (gnn_dagger.py:194-201): returns (steps done, final x)."""
    x = np.array(x0, dtype=np.float64)
    h = _flock.helpers(x, params)
    state = PortState(h['values'], h['network'], k, None, with_curr_gso)
    n = x.shape[0]
    with torch.no_grad():
        for _ in range(n_steps):
            mu = actor_forward(state.delay_state, state.delay_gso, weights, biases, 0, k)
            action = mu.permute(0, 1, 3, 2).reshape(n, -1).numpy()
            x, vals, net, _r = _flock.step(x, action, params)
            state = PortState(vals, net, k, state, with_curr_gso)
    return n_steps, x


def batched_step_time(B, N, K, F_, weights, biases, iters, seed=0):
    """Batched variant (B episodes per ATen call) of state update + actor forward on synthetic (S,X):
    the best case for the CPU path.  Returns seconds per batched step (env excluded)."""
    import time
    from . import synth
    g = torch.Generator().manual_seed(seed)
    A = torch.from_numpy(synth.make_adjacency_batch(seed, min(B, 8), N)).repeat((B + 7) // 8, 1, 1)[:B]
    Gp = torch.zeros(B, K, N, N)
    Gp[:, 0] = torch.eye(N)
    Xp = torch.randn(B, K, F_, N, generator=g)
    Xt = torch.randn(B, F_, N, generator=g)
    with torch.no_grad():
        def one():
            G = torch.zeros(B, K, N, N)
            G[:, 0] = torch.eye(N)
            G[:, 1:K] = torch.matmul(A[:, None], Gp[:, 0:K - 1])
            Xd = torch.zeros(B, K, F_, N)
            Xd[:, 0] = Xt
            Xd[:, 1:K] = Xp[:, 0:K - 1]
            return actor_forward(Xd, G, weights, biases, 0, K)
        one()
        t0 = time.perf_counter()
        for _ in range(iters):
            one()
        return (time.perf_counter() - t0) / iters
