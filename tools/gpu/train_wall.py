#!/usr/bin/env python3
"""Wall time of a whole DAGGER training run on the reference's own schedule (cfg/dagger.cfg: FlockingRelative-v0, N = 100, K = 3,
hidden [32, 32], 400 training episodes of 500 steps, 200 updates of 20 samples per episode, beta_coeff 0.993, lr 5e-5, 20 test
episodes) through this package's vectorised loop (learner/vec_dagger.py::train_dagger_vec, 64 episodes side by side per round):
where the seconds go (reset sampling on the host | collection launches | update rounds | test episodes) and what the trained
policy scores next to the expert and to idle agents.  One JSON record -> profiles/r04_train_wall.json.

    python tools/gpu/train_wall.py [--agents 100] [--taps 3] [--n-envs 64] [--lr 5e-5]
"""
import argparse
import configparser
import json
import os
import random
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tools'))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--agents', type=int, default=100)
    ap.add_argument('--taps', type=int, default=3)
    ap.add_argument('--n-envs', type=int, default=64)
    ap.add_argument('--lr', type=float, default=5e-5)
    ap.add_argument('--episodes', type=int, default=400)
    ap.add_argument('--updates-per-step', type=int, default=200)
    a = ap.parse_args()
    from multiagent_gnn_policies_amd.learner import vec_dagger as vd
    from train_policies import scripted_reward
    cp = configparser.ConfigParser()
    cp['DEFAULT'] = dict(alg='dagger_vec', batch_size='20', buffer_size='10000', updates_per_step=str(a.updates_per_step), seed='11',
                         actor_lr=repr(a.lr), n_train_episodes=str(a.episodes), beta_coeff='0.993', test_interval='40',
                         n_test_episodes='20', k=str(a.taps), hidden_size='32', n_layers='2', gamma='0.99', tau='0.5',
                         env='FlockingRelative-v0', v_max='3.0', comm_radius='1.0', n_agents=str(a.agents), n_actions='2',
                         n_states='6', debug='False', dt='0.01')
    cp['t'] = {}
    dev = torch.device('cuda:0')
    torch.zeros(1, device=dev)
    spent = {'collect_round (host reset sampling + collection launches)': 0.0, 'update rounds': 0.0, 'test episodes': 0.0}   # 'update rounds' include the next round's reset sampling, drawn under them
    counts = {'rounds': 0, 'updates': 0}

    def timed(fn, key, sync=True):
        def wrapped(*args, **kw):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            out = fn(*args, **kw)
            if sync:
                torch.cuda.synchronize()
            spent[key] += time.perf_counter() - t0
            return out
        return wrapped
    vd.collect_round = timed(vd.collect_round, 'collect_round (host reset sampling + collection launches)')
    vd.evaluate = timed(vd.evaluate, 'test episodes')
    # an update round = from the entry of FrameUpdates.run_sampled to the learner's end_updates() behind the loss read-out: the
    # host enqueues the round, draws the NEXT round's reset states while the GPU works (train_dagger_vec), then reads the loss
    orig_run = vd.FrameUpdates.run_sampled
    t_round = [None]

    def run_sampled(self, U, sampler=None):
        torch.cuda.synchronize()
        t_round[0] = time.perf_counter()
        counts['rounds'] += 1; counts['updates'] += U
        return orig_run(self, U, sampler)
    vd.FrameUpdates.run_sampled = run_sampled
    orig_end = vd.DAGGER.end_updates

    def end_updates(self):
        if t_round[0] is not None:
            torch.cuda.synchronize()
            spent['update rounds'] += time.perf_counter() - t_round[0]
            t_round[0] = None
        return orig_end(self)
    vd.DAGGER.end_updates = end_updates
    random.seed(11); np.random.seed(11); torch.manual_seed(11)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    res = vd.train_dagger_vec(cp['t'], dev, n_envs=a.n_envs)
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    from multiagent_gnn_policies_amd.learner.vec_dagger import _params_from_args
    p = _params_from_args(cp['t'])
    out = {
        "what": "one DAGGER training run on the reference's schedule (cfg/dagger.cfg) through train_dagger_vec",
        "shape": {"env": "FlockingRelative-v0", "agents": a.agents, "taps": a.taps, "hidden": [32, 32]},
        "schedule": {"train_episodes": a.episodes, "steps_per_episode": p.max_episode_steps, "updates_per_episode": a.updates_per_step,
                     "batch_size": 20, "lr": a.lr, "episodes_side_by_side": a.n_envs, "test_episodes": 20},
        "wall_s": wall,
        "wall_s_by_part": spent,
        "note_parts": "update rounds: enqueue + the next round's reset states drawn on the host while the GPU runs the updates + loss read-out",
        "wall_s_unaccounted": wall - sum(spent.values()),
        "rounds": counts['rounds'], "updates": counts['updates'],
        "us_per_update_incl_round_overheads": 1e6 * spent['update rounds'] / max(counts['updates'], 1),
        "env_steps_collected": res['updates'] and (counts['rounds'] * a.n_envs * p.max_episode_steps),
        "collect": res['collect'], "replay_bytes_per_transition": res.get('replay_bytes_per_transition'),
        "reward_trained_policy_mean_std": [res['mean'], res['std']],
        "reward_expert": scripted_reward(p, dev, p.max_episode_steps, 'expert'),
        "reward_idle_agents": scripted_reward(p, dev, p.max_episode_steps, 'idle'),
        "note": "the reference's loop on this host's CPU: 1.1 ms per update (oracle port, 1 thread: profiles/r04_dagger_update.json) "
                "x 80,000 updates = 88 s for the updates alone, 0.46 ms per env step x 200,000 steps = 92 s for the rollouts",
    }
    print(json.dumps(out))


if __name__ == '__main__':
    main()
