"""Spec + numpy oracle of the VECTORISED DAGGER data collection (TEST ORACLE; never imported by the product).

The reference collects with ONE environment (gnn_dagger.py:146-178).  This package's batched collection
(learner/vec_dagger.py, csrc/rollout.hip `rollout_kernel<.., CL = true>`) keeps the per-episode semantics and fixes what
batching leaves open:

  * beta of GLOBAL episode e is the reference's running product (gnn_dagger.py:148; oracle/imitation.py)
  * the per-step coin (gnn_dagger.py:157: np.random.binomial(1, beta)) is a COUNTER-BASED hash, so that a step's draw depends
    only on (seed, global episode index, steps since that episode's reset) -- not on lane, chunking or rank:

        coin(seed, episode, step) = fmix32( fmix32(seed + 0x9E3779B1 * episode)  ^  (0x85EBCA77 * step + 0xC2B2AE3D) )     (mod 2^32)
        the expert drives the step  <=>  coin < floor(beta * 2^32)          (beta >= 1: always; beta <= 0: never)

    with fmix32 = MurmurHash3's 32-bit finaliser (also used by the link-fading spec, oracle/flock.py).
  * every step files (state before the step, expert action for that state) -- gnn_dagger.py:174-178 -- as a compact FRAME:
    features x_t (6,N) fp32, membership bits of the network A_t, label (2,N) fp32, age (steps since reset).  The K-tap training
    state of a frame is state_with_delay.py:44-53 applied to the stored history: `state_from_history` below.
"""
import numpy as np

M32 = 0xFFFFFFFF


def fmix32(h):
    h &= M32
    h ^= h >> 16
    h = (h * 0x85EBCA6B) & M32
    h ^= h >> 13
    h = (h * 0xC2B2AE35) & M32
    h ^= h >> 16
    return h


def dagger_coin(seed, episode, step):
    a = fmix32((int(seed) + 0x9E3779B1 * int(episode)) & M32)
    return fmix32(a ^ ((0x85EBCA77 * int(step) + 0xC2B2AE3D) & M32))


def coin_threshold(beta):
    """floor(beta * 2^32) computed from the fp32 value of beta (the device holds beta as fp32), saturated to [0, 2^32]."""
    q = np.floor(float(np.float32(beta)) * 4294967296.0)
    return int(min(max(q, 0.0), 4294967296.0))


def expert_drives(seed, episode, step, beta):
    return dagger_coin(seed, episode, step) < coin_threshold(beta)


def state_from_history(feats, nets, age, K, dtype=np.float64):
    """feats[q] (6,N), nets[q] (N,N) for q = 0 (the frame's own state, time t) .. K-1 (time t-q); `age` = steps since reset of
    the frame's state.  Returns (delay_state (K,6,N), delay_gso (K,N,N)) as the reference builds them over the episode:
    delay_state[k] = x_{t-k} (0 for k > age); delay_gso[0] = I, delay_gso[j] = A_t A_{t-1} .. A_{t-j+1} (0 for j > age)."""
    F, N = feats[0].shape
    X = np.zeros((K, F, N), dtype=dtype)
    G = np.zeros((K, N, N), dtype=dtype)
    G[0] = np.eye(N, dtype=dtype)
    for k in range(K):
        if age >= k:
            X[k] = feats[k]
    for j in range(1, K):
        if age >= j:
            P = np.asarray(nets[0], dtype=dtype)
            for q in range(1, j):
                P = P @ np.asarray(nets[q], dtype=dtype)
            G[j] = P
    return X, G
