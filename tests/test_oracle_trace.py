"""CPU: the oracle's restatement of the DAGGER training loop (oracle/imitation.py) against the trace the REFERENCE's own
`train_dagger` left on the same fake environment (tests/golden/train_dagger_trace.npz, made by gen_golden.py in the build
container).  Row a9 of SURVEY.md section 8: beta schedule, coin flip per step, label transpose, `curr_size > batch_size`
gate, `updates_per_step` updates per episode, evaluation cadence, ring + random.sample (gnn_dagger.py:126-243,
replay_buffer.py:21-41)."""
import numpy as np

from conftest import load_golden, golden_weights
import fake_env
import trace_tools as tt
from oracle import imitation as oim


def run_oracle_trace(args=None, seed_env=None):
    g = load_golden('train_dagger_trace')
    args = args or tt.trace_args()
    Ws, bs = golden_weights(g, prefix='w0__')
    tr = tt.Trace()
    env = tt.RecordingEnv(fake_env.FakeFlockEnv(args.getint('n_agents'), episode_steps=int(g['episode_steps']),
                                                seed=args.getint('seed') if seed_env is None else seed_env), tr)
    Learner = tt.recording_learner(oim.DAGGER, tr, lambda l: {k: v.copy() for k, v in l.state_dict().items()})
    Replay = tt.recording_replay(oim.ReplayBuffer, tr)
    tt.seed_all(args.getint('seed'))
    with tt.recording_binomial(tr), tt.capture_stdout(tr):
        tr.stats = oim.train_dagger(env, args, lambda: Learner(args, Ws, bs), replay_cls=Replay)
    tr.final_weights = {k: v.copy() for k, v in tr.learner.state_dict().items()}
    return tr, g


def test_golden_trace_is_the_configuration_the_tests_assume():
    g = load_golden('train_dagger_trace')
    for k, v in tt.TRACE_CFG.items():
        assert str(g['cfg__' + k]) == v, "tests/trace_tools.TRACE_CFG drifted from the cfg the golden was recorded with"
    assert int(g['episode_steps']) == tt.TRACE_EPISODE_STEPS
    ev = str(g['events'])
    n_train, T = int(tt.TRACE_CFG['n_train_episodes']), tt.TRACE_EPISODE_STEPS
    assert ev.count('B') == n_train * T == len(g['binom_out'])               # one coin flip per training step (:157)
    assert ev.count('G') == ev.count('M') == len(g['losses']) == 15        # 3 updates after each of the 5 episodes
    assert ev.endswith('X')                                                # env.close() last (:242)
    # the ring wrapped: 40 inserts into 20 slots, positions 0..19 twice, size saturates at 20 (replay_buffer.py:27-32)
    assert list(g['insert_pos']) == list(range(20)) * 2
    assert list(g['insert_size']) == list(range(1, 21)) + [20] * 20
    # beta: running product 0.7, then clamped at 0.5 (gnn_dagger.py:148)
    assert np.array_equal(np.unique(g['binom_p']), np.array([0.5, 0.7]))
    assert 0 < g['step_expert_applied'][:n_train * T].sum() < n_train * T   # both the expert and the policy drove steps


def test_oracle_loop_reproduces_the_reference_trace():
    tr, g = run_oracle_trace()
    tt.compare_with_golden(tr, g)                     # asserts event order, betas, outcomes, indices, losses, weights


def test_trace_comparison_is_sensitive():
    """The comparison must notice a different loop: another env seed changes observations, hence labels and losses."""
    import pytest
    tr, g = run_oracle_trace(seed_env=4)
    with pytest.raises(AssertionError):
        tt.compare_with_golden(tr, g)
