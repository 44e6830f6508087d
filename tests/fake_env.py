"""A deterministic duck-typed environment with the call-site contract the reference's loops expect from gym_flock
(reference gnn_dagger.py:150,156,163,242): `env.reset()`, `env.env.controller()`, `env.step(u)`,
`env.close()`.  TEST INFRASTRUCTURE, own code: `tests/golden/gen_golden.py` drives the REFERENCE's `train_dagger` with it
(in the build container) to record a trace, and the parity tests drive the oracle loop and this package's loop with the
very same object, so every difference in the traces is a difference in the loop, not in the environment.

Design: the network sequence is pre-drawn per (episode, step) from the env's own RandomState and does NOT depend on the
actions (no radius threshold that fp32 / fp64 noise could flip); the observed values and the reward depend SMOOTHLY on the
applied actions through an accumulated offset q, so a 1e-6 difference in a policy action moves later observations by 1e-7,
never discontinuously.  The env never touches numpy's global RNG (the reference's beta coin flips own that stream,
gnn_dagger.py:157) nor Python's `random` (replay sampling, replay_buffer.py:40).
"""
import numpy as np


def _geometric_network(rs, n, mean_degree=4.0):
    """Row-normalised radius graph of n uniform points, zero diagonal (state_with_delay.py:26), float64."""
    side = np.sqrt(n * np.pi / mean_degree)
    p = rs.uniform(0.0, side, size=(n, 2))
    d = p[:, None, :] - p[None, :, :]
    r2 = d[:, :, 0] ** 2 + d[:, :, 1] ** 2
    np.fill_diagonal(r2, np.inf)
    adj = (r2 < 1.0).astype(np.float64)
    deg = adj.sum(axis=1, keepdims=True)
    return adj / np.maximum(deg, 1.0)


class FakeFlockEnv(object):
    """`env.env` is the env itself (the reference reaches the raw env through the TimeLimit wrapper's `.env`)."""

    def __init__(self, n_agents, n_states=6, n_actions=2, episode_steps=8, seed=0):
        self.n, self.f, self.na, self.T = n_agents, n_states, n_actions, episode_steps
        self.seed0 = seed
        self.episode = -1
        self.env = self
        self.closed = False
        self.log = []                       # ('reset', episode) / ('step', t, reward)

    # -- the gym-style surface ---------------------------------------------------------------------------------------
    def reset(self):
        self.episode += 1
        self.t = 0
        self.rs = np.random.RandomState(self.seed0 + 7919 * self.episode)
        self.base = self.rs.randn(self.T + 1, self.n, self.f)
        self.nets = [_geometric_network(self.rs, self.n) for _ in range(self.T + 1)]
        self.q = 0.1 * self.rs.randn(self.n, self.na)
        self.log.append(('reset', self.episode))
        return self._obs()

    def controller(self, centralized=None):
        """Expert action (N,nA) float64: a smooth function of the current observation."""
        v = self._values()
        return -0.5 * v[:, 0:self.na] + 0.25 * np.sin(v[:, self.na:2 * self.na])

    def step(self, action):
        a = np.asarray(action, dtype=np.float64)
        assert a.shape == (self.n, self.na)
        self.q = 0.9 * self.q + 0.1 * np.tanh(a)
        self.t += 1
        reward = -float(np.mean(self.q * self.q))
        done = self.t >= self.T
        self.log.append(('step', self.t, reward))
        return self._obs(), reward, done, {}

    def close(self):
        self.closed = True

    def seed(self, s=None):
        return [s]

    # -- internals ---------------------------------------------------------------------------------------------------
    def _values(self):
        v = 0.5 * self.base[self.t].copy()
        v[:, 0:self.na] += self.q
        v[:, self.na:2 * self.na] += 0.5 * self.q * self.base[self.t][:, 0:self.na]
        return v

    def _obs(self):
        return self._values(), self.nets[self.t]
