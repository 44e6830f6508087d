"""Replay memory with the reference's interface and semantics (reference learner/replay_buffer.py:4-49):
`Transition(state, action, done, next_state, reward)`; `ReplayBuffer(max_size)` with `insert`, `sample`, `clear` and the
public fields `buffer`, `max_size`, `curr_size`, `position`.  A full buffer overwrites its oldest entry; `sample` draws
without replacement from Python's global `random` stream (so seeding `random` reproduces the reference's batches).
"""
import random
from typing import NamedTuple, Any


class Transition(NamedTuple):
    state: Any
    action: Any
    done: Any
    next_state: Any
    reward: Any


class ReplayBuffer:

    def __init__(self, max_size=1000):
        self.max_size = max_size
        self.clear()

    def clear(self):
        self.buffer, self.curr_size, self.position = [], 0, 0

    def insert(self, sample):
        item = Transition(*sample)
        if len(self.buffer) < self.max_size:          # still growing: slot == position == len(buffer)
            self.buffer.append(item)
        else:                                         # ring: replace the oldest
            self.buffer[self.position] = item
        self.curr_size = len(self.buffer)
        self.position = (self.position + 1) % self.max_size

    def sample(self, num_samples):
        return random.sample(self.buffer, num_samples)
