"""Expert-controller baseline: no network, just `env.env.controller(centralized)` for `n_test_episodes` episodes;
reports mean / std of the episode reward.  Behaviour of reference learner/gnn_baseline.py:4-27."""
from .rollouts import reward_stats, run_episode


def train_baseline(env, args):
    centralized = args.getboolean('centralized')
    rewards = [run_episode(env, lambda _obs: env.env.controller(centralized))
               for _ in range(args.getint('n_test_episodes'))]
    env.close()
    return reward_stats(rewards)
