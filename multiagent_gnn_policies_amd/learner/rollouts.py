"""Episode-rollout helpers shared by the DAGGER / cloning / baseline loops (gym-style env, one episode at a time)."""
import numpy as np

from .state_with_delay import MultiAgentStateWithDelay


def run_episode(env, act, on_step=None):
    """Roll one episode: `act(obs)` -> action array; returns the summed reward.
    `on_step(obs, action, next_obs, reward, done)` is called after every env.step (used to fill replay memories)."""
    obs = env.reset()
    total, done = 0.0, False
    while not done:
        action = act(obs)
        next_obs, reward, done, _ = env.step(action)
        if on_step is not None:
            on_step(obs, action, next_obs, reward, done)
        total += reward
        obs = next_obs
    return total


def reward_stats(rewards):
    return {'mean': np.mean(rewards), 'std': np.std(rewards)}


class PolicyRunner(object):
    """Carries the delayed state of ONE environment across steps so a learner can be used as `act(obs)`."""

    def __init__(self, learner, device, args):
        self.learner, self.device, self.args = learner, device, args
        self.state = None
        self._fresh = True

    def reset(self):
        self.state = None

    def observe(self, obs):
        """Fold a new observation into the delay line (reference gnn_dagger.py:150,165)."""
        self.state = MultiAgentStateWithDelay(self.device, self.args, obs, prev_state=self.state)
        return self.state

    def act(self, obs):
        return self.learner.select_action(self.observe(obs)).cpu().numpy()


def policy_episode_reward(env, learner, device, args):
    """One policy-only episode (the reference's test loop, gnn_dagger.py:194-203)."""
    runner = PolicyRunner(learner, device, args)
    return run_episode(env, runner.act)
