"""Behaviour cloning (reference learner/gnn_cloning.py:123-213): like DAGGER but the environment is
always stepped with the expert action, evaluation every `test_interval` episodes keeps the best
mean reward (and saves that model when `debug` and `fname` are set).  The learner is the same network
and update as DAGGER's (the reference duplicates the class, gnn_cloning.py:17-120)."""
import numpy as np
import torch

from .gnn_dagger import DAGGER
from .rollouts import policy_episode_rewards
from .replay_buffer import ReplayBuffer, Transition
from .state_with_delay import MultiAgentStateWithDelay

ImitationLearning = DAGGER


def train_cloning(env, args, device):
    debug = args.getboolean('debug')
    memory = ReplayBuffer(max_size=args.getint('buffer_size'))
    learner = ImitationLearning(device, args)

    n_a = args.getint('n_actions')
    n_agents = args.getint('n_agents')
    batch_size = args.getint('batch_size')
    updates_per_step = args.getint('updates_per_step')
    n_train_episodes = args.getint('n_train_episodes')
    test_interval = args.getint('test_interval')
    n_test_episodes = args.getint('n_test_episodes')

    total_numsteps = 0
    updates = 0
    stats = {'mean': -1.0 * np.inf, 'std': 0}

    for i in range(n_train_episodes):
        state = MultiAgentStateWithDelay(device, args, env.reset(), prev_state=None)
        done = False
        policy_loss_sum = 0
        while not done:
            optimal_action = env.env.controller()
            next_obs, reward, done, _ = env.step(optimal_action)
            next_state = MultiAgentStateWithDelay(device, args, next_obs, prev_state=state)
            total_numsteps += 1
            notdone = torch.tensor([float(not done)], device=device)
            reward_t = torch.tensor([float(reward)], device=device)
            label = torch.from_numpy(np.ascontiguousarray(np.asarray(optimal_action, dtype=np.float32).T))
            label = label.reshape((1, 1, n_a, n_agents)).to(device)
            memory.insert(Transition(state, label, notdone, next_state, reward_t))
            state = next_state

        if memory.curr_size > batch_size:
            for _ in range(updates_per_step):
                batch = Transition(*zip(*memory.sample(batch_size)))
                policy_loss_sum += learner.gradient_step(batch)
                updates += 1

        if i % test_interval == 0:
            test_rewards = policy_episode_rewards(env, learner, device, args, n_test_episodes)
            mean_reward = np.mean(test_rewards)
            if stats['mean'] < mean_reward:
                stats['mean'] = mean_reward
                stats['std'] = np.std(test_rewards)
                if debug and args.get('fname'):
                    learner.save_model(args.get('env'), suffix=args.get('fname'))
            if debug:
                print("Episode: {}, updates: {}, total numsteps: {}, reward: {}, policy loss: {}".format(
                    i, updates, total_numsteps, mean_reward, policy_loss_sum))

    env.close()
    return stats
