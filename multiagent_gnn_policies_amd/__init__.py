"""MI355X-native hot path of katetolstaya/multiagent_gnn_policies.

Host side: Python on PyTorch-ROCm mirroring the reference's `Actor` nn.Module, state builder,
DAGGER learner and gym-style flocking environment.  All hot arithmetic: hand-written HIP kernels for
gfx950 in libmgp.so, reached through the C ABI of include/mgp.h (ctypes, raw device pointers, the
current torch HIP stream).  There is no CPU compute path in this package.
"""
from ._lib import MgpError, lib  # noqa: F401

__version__ = '0.1.0'
