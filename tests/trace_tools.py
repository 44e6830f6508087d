"""Records what a `train_dagger` run DOES -- the control flow the reference spells out in gnn_dagger.py:126-243 and
replay_buffer.py:21-41 -- independently of whose loop is running: the reference's (tests/golden/gen_golden.py, build
container only), the oracle's restatement (oracle/imitation.py) or this package's (`learner/imitation.py`).  TEST
INFRASTRUCTURE.  The recorder hooks the three seams every implementation shares:

  * the environment (tests/fake_env.FakeFlockEnv): reset / controller / step calls, applied actions, rewards
  * the replay memory class: insert position + label + size after insert; indices drawn by every `sample`
  * the learner class: `select_action` outputs and `gradient_step` losses
  * `np.random.binomial` (the beta coin flip, gnn_dagger.py:157): probability passed in and outcome

and keeps one event string (one character per event, in call order) so that the ORDER of all of it is pinned too.
"""
import contextlib
import io

import numpy as np

EVENTS = dict(reset='R', controller='C', binomial='B', select='A', step='S', insert='I', sample='M', update='G', close='X')


class Trace(object):
    def __init__(self):
        self.events = []
        self.binom_p, self.binom_out = [], []
        self.step_actions, self.step_rewards, self.step_done = [], [], []
        self.step_expert_applied = []          # 1 when the action handed to env.step IS the last controller() output
        self.insert_pos, self.insert_size, self.insert_labels = [], [], []
        self.sample_idx = []
        self.losses = []
        self.select_out = []
        self.printed = ''
        self.stats = None
        self.final_weights = None
        self.initial_weights = None

    def to_npz_dict(self):
        d = dict(events=np.array(''.join(self.events)),
                 binom_p=np.array(self.binom_p, dtype=np.float64), binom_out=np.array(self.binom_out, dtype=np.int64),
                 step_actions=np.array(self.step_actions, dtype=np.float64),
                 step_rewards=np.array(self.step_rewards, dtype=np.float64),
                 step_done=np.array(self.step_done, dtype=np.int64),
                 step_expert_applied=np.array(self.step_expert_applied, dtype=np.int64),
                 insert_pos=np.array(self.insert_pos, dtype=np.int64), insert_size=np.array(self.insert_size, dtype=np.int64),
                 insert_labels=np.array(self.insert_labels, dtype=np.float32),
                 sample_idx=np.array(self.sample_idx, dtype=np.int64), losses=np.array(self.losses, dtype=np.float64),
                 select_out=np.array(self.select_out, dtype=np.float32), printed=np.array(self.printed),
                 stats_mean=np.float64(self.stats['mean']), stats_std=np.float64(self.stats['std']))
        for k, v in (self.initial_weights or {}).items():
            d['w0__' + k] = v
        for k, v in (self.final_weights or {}).items():
            d['wF__' + k] = v
        return d


def _np(a):
    if hasattr(a, 'detach'):
        a = a.detach().cpu().numpy()
    return np.asarray(a)


class RecordingEnv(object):
    """Wraps a FakeFlockEnv; `.env` is the wrapper itself so that `env.env.controller()` is recorded too."""

    def __init__(self, inner, trace):
        self._inner, self._tr = inner, trace
        self.env = self
        self._last_expert = None

    def reset(self):
        self._tr.events.append(EVENTS['reset'])
        return self._inner.reset()

    def controller(self, *a, **kw):
        self._tr.events.append(EVENTS['controller'])
        self._last_expert = self._inner.controller(*a, **kw)
        return self._last_expert

    def step(self, action):
        tr = self._tr
        tr.events.append(EVENTS['step'])
        a = _np(action)
        tr.step_actions.append(a.astype(np.float64))
        tr.step_expert_applied.append(int(action is self._last_expert))
        out = self._inner.step(action)
        tr.step_rewards.append(float(out[1]))
        tr.step_done.append(int(bool(out[2])))
        return out

    def close(self):
        self._tr.events.append(EVENTS['close'])
        return self._inner.close()

    def seed(self, s=None):
        return self._inner.seed(s)

    def __getattr__(self, name):
        return getattr(self._inner, name)


def recording_replay(base_cls, trace):
    """Subclass of a ReplayBuffer class (reference's or this package's) that logs inserts and sampled indices."""

    class RecordingReplay(base_cls):
        def insert(self, sample):
            pos = self.position
            base_cls.insert(self, sample)
            trace.events.append(EVENTS['insert'])
            trace.insert_pos.append(pos)
            trace.insert_size.append(self.curr_size)
            trace.insert_labels.append(_np(self.buffer[pos].action).astype(np.float32))

        def sample(self, num_samples):
            got = base_cls.sample(self, num_samples)
            where = {id(t): i for i, t in enumerate(self.buffer)}
            trace.events.append(EVENTS['sample'])
            trace.sample_idx.append([where[id(t)] for t in got])
            return got

    return RecordingReplay


def recording_learner(base_cls, trace, state_dict_of):
    """Subclass of a DAGGER class that logs select_action outputs and gradient_step losses, and snapshots the initial
    weights.  `state_dict_of(learner)` -> {name: ndarray}."""

    class RecordingLearner(base_cls):
        def __init__(self, *a, **kw):
            base_cls.__init__(self, *a, **kw)
            trace.initial_weights = state_dict_of(self)
            trace.learner = self

        def select_action(self, state):
            out = base_cls.select_action(self, state)
            trace.events.append(EVENTS['select'])
            trace.select_out.append(_np(out).astype(np.float32))
            return out

        def gradient_step(self, batch):
            loss = base_cls.gradient_step(self, batch)
            trace.events.append(EVENTS['update'])
            trace.losses.append(float(loss))
            return loss

    return RecordingLearner


@contextlib.contextmanager
def recording_binomial(trace):
    """Patch np.random.binomial (global stream: what the reference calls at gnn_dagger.py:157) with a logging wrapper."""
    orig = np.random.binomial

    def wrapped(n, p, *a, **kw):
        out = orig(n, p, *a, **kw)
        trace.events.append(EVENTS['binomial'])
        trace.binom_p.append(float(p))
        trace.binom_out.append(int(out))
        return out
    np.random.binomial = wrapped
    try:
        yield
    finally:
        np.random.binomial = orig


@contextlib.contextmanager
def capture_stdout(trace):
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        yield
    trace.printed = buf.getvalue()


TRACE_CFG = dict(alg='dagger', batch_size='5', buffer_size='20', updates_per_step='3', seed='3', actor_lr='1e-3',
                 n_train_episodes='5', beta_coeff='0.7', test_interval='2', n_test_episodes='2', k='3', hidden_size='16',
                 gamma='0.99', tau='0.5', env='FakeFlock-v0', v_max='3.0', comm_radius='1.0', n_agents='12',
                 n_actions='2', n_states='6', debug='True', dt='0.01')
TRACE_EPISODE_STEPS = 8


def trace_args(**kw):
    import configparser
    cp = configparser.ConfigParser()
    base = dict(TRACE_CFG)
    base.update({k: str(v) for k, v in kw.items()})
    cp['DEFAULT'] = base
    cp['test'] = {}
    return cp['test']


def seed_all(seed):
    """The three global streams reference train.py:26-28 seeds (the env's own stream is FakeFlockEnv's seed argument)."""
    import random
    import torch
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)


def parse_printed(text):
    """[(episode, updates, total_numsteps, reward, policy_loss)] from the loop's debug lines (gnn_dagger.py:213-219)."""
    import re
    rows = []
    for line in text.strip().splitlines():
        m = re.match(r'Episode: (\d+), updates: (\d+), total numsteps: (\d+), reward: (\S+), policy loss: (\S+)$', line.strip())
        assert m, "unexpected line printed by the loop: %r" % line
        rows.append((int(m.group(1)), int(m.group(2)), int(m.group(3)), float(m.group(4)), float(m.group(5))))
    return rows


def compare_with_golden(tr, g, loss_tol=1e-5, weight_tol=2e-6, action_tol=1e-5):
    """Assert that the recorded run `tr` did what the reference did (golden dict `g` from train_dagger_trace.npz):
    IDENTICAL control flow (event order, beta handed to every coin flip, outcomes, who drove each step, ring positions,
    buffer sizes, minibatch indices, printed episode / update / step counters) and values within the fp32 tolerances."""
    got = tr.to_npz_dict()
    assert str(got['events']) == str(g['events']), "event order differs from the reference's loop"
    assert np.array_equal(got['binom_p'], g['binom_p']), "beta schedule differs (gnn_dagger.py:148 is a running product)"
    assert np.array_equal(got['binom_out'], g['binom_out'])
    assert np.array_equal(got['step_expert_applied'], g['step_expert_applied'])
    assert np.array_equal(got['step_done'], g['step_done'])
    assert np.array_equal(got['insert_pos'], g['insert_pos']) and np.array_equal(got['insert_size'], g['insert_size'])
    assert np.array_equal(got['sample_idx'], g['sample_idx']), "minibatch indices differ (replay_buffer.py:40)"
    assert got['insert_labels'].shape == g['insert_labels'].shape == (len(g['insert_pos']), 1, 1, 2, int(g['cfg__n_agents']))
    err = {}
    err['labels'] = float(np.max(np.abs(got['insert_labels'] - g['insert_labels'])))
    err['actions'] = float(np.max(np.abs(got['step_actions'] - g['step_actions'])))
    err['select'] = float(np.max(np.abs(got['select_out'] - g['select_out'])))
    err['rewards'] = float(np.max(np.abs(got['step_rewards'] - g['step_rewards'])))
    err['losses'] = float(np.max(np.abs(got['losses'] - g['losses'])))
    assert err['labels'] <= action_tol and err['actions'] <= action_tol and err['select'] <= action_tol, err
    assert err['rewards'] <= action_tol, err
    assert err['losses'] <= loss_tol, err
    werr = 0.0
    for k in g:
        if k.startswith('wF__'):
            werr = max(werr, float(np.max(np.abs(got[k] - g[k]))))
        if k.startswith('w0__'):
            assert np.array_equal(got[k], g[k]), "initial weights differ: " + k
    err['weights'] = werr
    assert werr <= weight_tol, err
    rows, ref_rows = parse_printed(str(got['printed'])), parse_printed(str(g['printed']))
    assert [r[:3] for r in rows] == [r[:3] for r in ref_rows], "printed episode / updates / numsteps differ"
    for r, q in zip(rows, ref_rows):
        assert abs(r[3] - q[3]) <= 1e-5 * max(1.0, abs(q[3])) and abs(r[4] - q[4]) <= 3 * loss_tol, (r, q)
    assert abs(float(got['stats_mean']) - float(g['stats_mean'])) <= 1e-5
    assert abs(float(got['stats_std']) - float(g['stats_std'])) <= 1e-5
    return err
