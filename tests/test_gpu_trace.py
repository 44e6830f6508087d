"""GPU: this package's `train_dagger` (learner/gnn_dagger.py -> learner/imitation.py::ImitationRun, HIP kernels underneath)
must DO what the reference's `train_dagger` does (gnn_dagger.py:126-243, replay_buffer.py:21-41): row a9 of SURVEY.md
section 8.  The golden trace was recorded from the reference's own loop on tests/fake_env.FakeFlockEnv
(tests/golden/gen_golden.py::gen_train_trace); the same environment, seeds and recorder are used here.  Control flow must be
IDENTICAL (event order, beta of every coin flip, outcomes, who drove each step, ring positions, minibatch indices, printed
counters); losses within 1e-5, weights after 15 Adam steps within 2e-6, actions / labels within 1e-5."""
import numpy as np
import pytest
import torch

from conftest import load_golden
import fake_env
import trace_tools as tt

pytestmark = pytest.mark.gpu


def _sd(learner):
    return {k.replace('.', '__'): v.detach().cpu().numpy().copy() for k, v in learner.actor.state_dict().items()}


def _run_product_trace(monkeypatch, graphed, train_step):
    from multiagent_gnn_policies_amd.learner import gnn_dagger as gd, imitation as im
    g = load_golden('train_dagger_trace')
    args = tt.trace_args()
    tr = tt.Trace()
    env = tt.RecordingEnv(fake_env.FakeFlockEnv(args.getint('n_agents'), episode_steps=int(g['episode_steps']),
                                                seed=args.getint('seed')), tr)
    base = gd.DAGGER

    class Learner(base):
        def __init__(self, *a, **kw):
            base.__init__(self, *a, **kw)
            self.use_graphed_update, self.use_train_step = graphed, train_step

    monkeypatch.setattr(gd, 'DAGGER', tt.recording_learner(Learner, tr, _sd))
    monkeypatch.setattr(im, 'ReplayBuffer', tt.recording_replay(im.ReplayBuffer, tr))
    tt.seed_all(args.getint('seed'))
    with tt.recording_binomial(tr), tt.capture_stdout(tr):
        tr.stats = gd.train_dagger(env, args, torch.device('cuda:0'))
    tr.final_weights = _sd(tr.learner)
    return tr, g


@pytest.mark.parametrize('graphed,train_step', [(True, True), (True, False), (False, True), (False, False)],
                         ids=['graph-2launch', 'graph-5launch', 'eager-train_grads', 'eager-autograd'])
def test_train_dagger_reproduces_the_reference_trace(monkeypatch, graphed, train_step):
    tr, g = _run_product_trace(monkeypatch, graphed, train_step)
    err = tt.compare_with_golden(tr, g, loss_tol=1e-5, weight_tol=2e-6, action_tol=1e-5)
    print('train_dagger trace vs reference:', err)
    assert tr.learner.actor_optim.step_count == len(g['losses'])
    assert int(tr.learner.actor_optim.step_dev.item()) == len(g['losses'])
