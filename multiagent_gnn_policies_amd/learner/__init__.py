from .actor import Actor  # noqa: F401
from .state_with_delay import MultiAgentStateWithDelay  # noqa: F401
from .replay_buffer import ReplayBuffer, Transition  # noqa: F401
