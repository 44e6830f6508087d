"""Behaviour cloning -- drop-in for reference learner/gnn_cloning.py:123-213.

The environment is always stepped with the expert action, the policy is evaluated every `test_interval` episodes, and
the best evaluation is what `train_cloning` returns (that model is saved when `debug` and `fname` are set).  Network and
update are DAGGER's (the reference duplicates the class, gnn_cloning.py:17-120); the loop itself is `ImitationRun`.
"""
from .gnn_dagger import DAGGER
from .imitation import ImitationRun

ImitationLearning = DAGGER


def train_cloning(env, args, device):
    run = ImitationRun(env, ImitationLearning(device, args), args, device)
    return run.run(lambda e: None, eval_always=True, keep_best=True)
