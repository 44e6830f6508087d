"""Expert-controller baseline (reference learner/gnn_baseline.py:4-27): no network, rolls out
`env.env.controller(centralized)` for `n_test_episodes` and reports mean/std of the episode reward."""
import numpy as np


def train_baseline(env, args):
    n_test_episodes = args.getint('n_test_episodes')
    centralized = args.getboolean('centralized')
    stats = {'mean': -1.0 * np.inf, 'std': 0}
    test_rewards = []
    for _ in range(n_test_episodes):
        ep_reward = 0
        env.reset()
        done = False
        while not done:
            action = env.env.controller(centralized)
            _, reward, done, _ = env.step(action)
            ep_reward += reward
        test_rewards.append(ep_reward)
    stats['mean'] = np.mean(test_rewards)
    stats['std'] = np.std(test_rewards)
    env.close()
    return stats
