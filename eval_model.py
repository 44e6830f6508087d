"""Checkpoint evaluation harness: roll a saved Actor out in an environment and report the episode rewards.

Counterpart of the reference's two evaluation scripts (test_model.py:14-47: print every episode reward of
`models/actor_FlockingRelative-v0_dagger_k3`; test_model_transfer.py:14-59,84-99: one checkpoint per `k`, print
`section, mean, std`), folded into one command:

    python3 eval_model.py cfg/flocking_dagger_n100_k3.cfg                      # per-episode rewards (test_model.py)
    python3 eval_model.py cfg/transfer.cfg --actor models/actor_X_transfer{k}   # {k} <- section's k, prints stats
    python3 eval_model.py cfg/... --lanes 256                                   # device-resident lanes of episodes

Checkpoints are `torch.save`d state_dicts with the reference's keys (`conv_layers.{i}.weight/.bias`) or the `.npz`
weight fixture under tests/golden/.  `--lanes B` evaluates B episodes at a time with the batched simulator (two kernel
launches per env step for all lanes); the default is the reference's one-environment gym-style loop.  No rendering.
"""
import argparse
import configparser
import os
import random

import numpy as np
import torch

from multiagent_gnn_policies_amd import envs

DEFAULT_ACTOR = os.path.join('tests', 'golden', 'ckpt_dagger_k3.npz')   # the reference's shipped K=3 / H=32 policy


def _seed(seed, env):
    env.seed(seed)
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)


def evaluate_section(args, actor_path, lanes=0, k=None, verbose=True):
    """Episode rewards (list of float) of one cfg section."""
    from multiagent_gnn_policies_amd.learner.gnn_dagger import DAGGER
    from multiagent_gnn_policies_amd.learner.rollouts import PolicyRunner, run_episode, fast_loop_mode
    if not torch.cuda.is_available():
        raise RuntimeError("eval_model.py needs an MI355X (HIP device); this framework has no CPU compute path")
    device = torch.device("cuda:0")
    env = envs.make(args.get('env'), device=str(device))
    if hasattr(env.env, 'params_from_cfg'):
        env.env.params_from_cfg(args)
    _seed(args.getint('seed'), env)
    learner = DAGGER(device, args, k=k)
    learner.load_model(actor_path, device)
    n_episodes = args.getint('n_test_episodes')
    if lanes:
        from multiagent_gnn_policies_amd.envs.flocking import VecFlock
        from multiagent_gnn_policies_amd.learner.state_with_delay import BatchedDelayState
        from multiagent_gnn_policies_amd.learner.vec_dagger import _params_from_args, evaluate
        p = _params_from_args(args)
        sim = VecFlock(lanes, p, device, with_expert=False)
        state = BatchedDelayState(device, lanes, learner.actor.k, args.getint('n_states'), p.n_agents)
        rewards = evaluate(learner, sim, state, n_episodes, p.max_episode_steps)
    else:
        rewards = []
        with fast_loop_mode(env):                                  # nothing crosses PCIe per step; previous mode restored
            for _ in range(n_episodes):
                runner = PolicyRunner(learner, device, args) if k is None else _RunnerK(learner, device, args, k)
                rewards.append(run_episode(env, runner.act))
                if verbose:
                    print(rewards[-1])
    env.close()
    return rewards


class _RunnerK(object):
    """PolicyRunner with the delay depth overridden (test_model_transfer.py passes k explicitly)."""

    def __init__(self, learner, device, args, k):
        from multiagent_gnn_policies_amd.learner.state_with_delay import MultiAgentStateWithDelay
        self._mk = lambda obs, prev: MultiAgentStateWithDelay(device, args, obs, prev_state=prev, k=k)
        self.learner, self.state = learner, None

    def act(self, obs):
        self.state = self._mk(obs, self.state)
        return self.learner.select_action(self.state).cpu().numpy()


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__.split('\n')[0])
    ap.add_argument('cfg')
    ap.add_argument('--actor', default=DEFAULT_ACTOR,
                    help="checkpoint path; '{k}' is replaced by the section's k (transfer evaluation)")
    ap.add_argument('--lanes', type=int, default=0, help="evaluate this many episodes at a time on the device")
    ap.add_argument('--stats', action='store_true', help="print `section, mean, std` instead of every episode reward")
    ns = ap.parse_args(argv)
    config = configparser.ConfigParser()
    if not config.read(ns.cfg):
        raise FileNotFoundError(ns.cfg)
    per_k = '{k}' in ns.actor
    stats_mode = ns.stats or per_k or ns.lanes > 0
    sections = config.sections() or [config.default_section]
    if config.sections() and config[sections[0]].get('header') is not None:
        print(config[sections[0]].get('header'))
    out = {}
    for name in sections:
        args = config[name]
        k = args.getint('k') if per_k else None
        rewards = evaluate_section(args, ns.actor.format(k=k) if per_k else ns.actor, lanes=ns.lanes, k=k,
                                   verbose=not stats_mode or args.getboolean('debug', fallback=False))
        out[name] = rewards
        if stats_mode:
            prefix = (name + ", ") if config.sections() else ""
            print(prefix + str(np.mean(rewards)) + ", " + str(np.std(rewards)))
    return out


if __name__ == "__main__":
    main()
