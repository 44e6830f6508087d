"""Replay memory with the reference's interface (reference learner/replay_buffer.py:4-49):
`Transition(state, action, done, next_state, reward)`, `ReplayBuffer(max_size).insert/sample/clear`,
ring overwrite of the oldest sample, `random.sample` without replacement from Python's global RNG.
"""
from collections import namedtuple
import random

Transition = namedtuple('Transition', ('state', 'action', 'done', 'next_state', 'reward'))


class ReplayBuffer(object):

    def __init__(self, max_size=1000):
        self.buffer = []
        self.max_size = max_size
        self.curr_size = 0
        self.position = 0

    def insert(self, sample):
        if self.curr_size < self.max_size:
            self.buffer.append(None)
            self.curr_size += 1
        self.buffer[self.position] = Transition(*sample)
        self.position = (self.position + 1) % self.max_size

    def sample(self, num_samples):
        return random.sample(self.buffer, num_samples)

    def clear(self):
        self.buffer = []
        self.curr_size = 0
        self.position = 0
