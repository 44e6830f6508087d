"""The one-environment imitation loop shared by `train_dagger` and `train_cloning`.

The reference writes this loop twice (learner/gnn_dagger.py:126-243 and learner/gnn_cloning.py:123-213); the two copies
differ in three places only, which are the three arguments of `ImitationRun.run`:

  who drives the environment   DAGGER: the expert with probability beta, else the policy (gnn_dagger.py:156-161);
                               cloning: always the expert (gnn_cloning.py:151-153)
  when it evaluates            DAGGER: every `test_interval` episodes if `debug`, plus once at the end (:190,:222-237);
                               cloning: every `test_interval` episodes, always (:182)
  what it reports              DAGGER: the final evaluation; cloning: the best evaluation, saving that model (:203-209)

Under torchrun the episodes are dealt round-robin to the ranks (global episode index = local index * world + rank, so
the beta schedule stays per global episode), every rank performs the same number of updates, and `DAGGER.gradient_step`
averages gradients with one flat all-reduce per update.  Evaluation episodes are sharded the same way and gathered.
"""
from dataclasses import dataclass

import numpy as np
import torch

from .. import parallel
from .replay_buffer import ReplayBuffer, Transition
from .rollouts import policy_episode_rewards
from .state_with_delay import MultiAgentStateWithDelay


@dataclass
class LoopSettings:
    """The cfg keys the loop reads (reference gnn_dagger.py:128-146), parsed once."""
    debug: bool
    buffer_size: int
    n_actions: int
    n_agents: int
    batch_size: int
    updates_per_step: int
    n_train_episodes: int
    test_interval: int
    n_test_episodes: int
    fname: str
    env_name: str

    @classmethod
    def from_args(cls, args):
        geti = args.getint
        return cls(debug=args.getboolean('debug'), buffer_size=geti('buffer_size'), n_actions=geti('n_actions'),
                   n_agents=geti('n_agents'), batch_size=geti('batch_size'), updates_per_step=geti('updates_per_step'),
                   n_train_episodes=geti('n_train_episodes'), test_interval=geti('test_interval'),
                   n_test_episodes=geti('n_test_episodes'), fname=args.get('fname'), env_name=args.get('env'))


class ImitationRun(object):
    def __init__(self, env, learner, args, device):
        self.env, self.learner, self.args, self.device = env, learner, args, device
        self.cfg = LoopSettings.from_args(args)
        self.memory = ReplayBuffer(max_size=self.cfg.buffer_size)
        self.rank, self.world = parallel.rank(), parallel.world_size()
        lo, hi = parallel.shard_range(self.cfg.n_test_episodes)
        self.n_test_local = max(1, hi - lo) if self.world > 1 else self.cfg.n_test_episodes
        self.total_numsteps = 0
        self.updates = 0
        from .rollouts import fast_loop_mode
        # this package's simulator: the one-environment loop never leaves the device; run() hands the env back in the mode
        # it came in (a gym_flock-style caller keeps getting numpy observations afterwards)
        self._mode = fast_loop_mode(env)
        self.fast = self._mode.__enter__()

    # ------------------------------------------------------------------ the three stages of one training episode
    def collect(self, beta):
        """One episode into the replay memory, every step labelled with the expert action (N,nA) -> (1,1,nA,N).
        `beta` None: the expert drives (no RNG draw); else the expert drives a step with probability beta."""
        c, env, dev = self.cfg, self.env, self.device
        state = MultiAgentStateWithDelay(dev, self.args, env.reset(), prev_state=None)
        done = False
        while not done:
            expert = env.env.controller()
            if beta is None or np.random.binomial(1, beta) > 0:
                applied = expert
            else:
                applied = self.learner.select_action(state)
                if not self.fast:
                    applied = applied.cpu().numpy()
            obs, reward, done, _ = env.step(applied)
            nxt = MultiAgentStateWithDelay(dev, self.args, obs, prev_state=state)
            if torch.is_tensor(expert):              # fast loop: (N,nA) fp32 on the device -> (1,1,nA,N), reward 0-d tensor
                label = expert.t().reshape((1, 1, c.n_actions, c.n_agents)).contiguous()
                rew = reward.to(torch.float32).reshape(1)
            else:
                label = torch.from_numpy(np.ascontiguousarray(np.asarray(expert, dtype=np.float32).T))
                label = label.reshape((1, 1, c.n_actions, c.n_agents)).to(dev)
                rew = torch.tensor([float(reward)], device=dev)
            self.memory.insert(Transition(state, label, self._notdone_flag(done), nxt, rew))
            state = nxt
            self.total_numsteps += 1

    def _notdone_flag(self, done):
        """The transition's `notdone` (reference gnn_dagger.py:167: torch.Tensor([not done])) as one of two device constants --
        a fresh host-to-device tensor per environment step cost 19 us of a 150 us step; nothing writes into it."""
        flags = getattr(self, '_notdone', None)
        if flags is None:
            flags = self._notdone = (torch.tensor([1.0], device=self.device), torch.tensor([0.0], device=self.device))
        return flags[1 if done else 0]

    def fit(self):
        """`updates_per_step` minibatch updates once the memory holds more than one batch; returns the loss sum."""
        c = self.cfg
        if self.memory.curr_size <= c.batch_size:
            return 0
        loss_sum = 0
        begin, end = getattr(self.learner, 'begin_updates', None), getattr(self.learner, 'end_updates', None)
        if begin is not None:
            begin()                                  # data-parallel runs: align the ranks before the exchanges
        for _ in range(c.updates_per_step):
            loss_sum += self.learner.gradient_step(Transition(*zip(*self.memory.sample(c.batch_size))))
        if end is not None:
            end()
        self.updates += c.updates_per_step
        return loss_sum

    def evaluate(self):
        """Policy-only episodes (this rank's share), gathered over the ranks."""
        r = policy_episode_rewards(self.env, self.learner, self.device, self.args, self.n_test_local)
        return parallel.all_gather_floats(r)

    def log(self, episode, rewards, loss_sum):
        if self.cfg.debug and self.rank == 0:
            print("Episode: {}, updates: {}, total numsteps: {}, reward: {}, policy loss: {}".format(
                episode, self.updates, self.total_numsteps * self.world, np.mean(rewards), loss_sum))

    def save(self):
        if self.cfg.debug and self.cfg.fname and self.rank == 0:
            self.learner.save_model(self.cfg.env_name, suffix=self.cfg.fname)

    # ------------------------------------------------------------------ the loop
    def run(self, beta_of_episode, eval_always, keep_best):
        try:
            return self._run(beta_of_episode, eval_always, keep_best)
        finally:
            self._mode.__exit__(None, None, None)

    def _run(self, beta_of_episode, eval_always, keep_best):
        c, world = self.cfg, self.world
        stats = {'mean': -1.0 * np.inf, 'std': 0}
        for i in range((c.n_train_episodes + world - 1) // world):
            self.collect(beta_of_episode(i * world + self.rank))
            loss_sum = self.fit()
            if (i * world) % c.test_interval < world and (eval_always or c.debug):
                rewards = self.evaluate()
                if keep_best and stats['mean'] < np.mean(rewards):
                    stats = {'mean': np.mean(rewards), 'std': np.std(rewards)}
                    self.save()
                self.log(i * world, rewards, loss_sum)
        if not keep_best:
            rewards = self.evaluate()
            stats = {'mean': np.mean(rewards), 'std': np.std(rewards)}
            self.save()
        self.env.close()
        return stats
