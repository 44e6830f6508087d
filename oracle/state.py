"""numpy restatement of the delayed-GSO / delay-line state update (TEST ORACLE).

Follows reference learner/state_with_delay.py:
  * fp64 -> fp32 cast of the env tuple, (N,F)->(1,1,F,N), (N,N)->(1,1,N,N)      :29-35
  * curr_gso[0] = I ; curr_gso[j] = A_t @ curr_gso[j-1]                          :38-41
  * delay_gso[0] = I ; delay_gso[1:K] = A_t @ prev.delay_gso[0:K-1] (0 w/o prev) :44-47
  * delay_state[0] = x_t ; delay_state[1:K] = prev.delay_state[0:K-1]            :50-53

Batched over a leading episode axis B (the reference always has B == 1).
"""
import numpy as np


def cast_env_state(values, network, dtype=np.float32):
    """(N,F) f64, (N,N) f64 -> (1,1,F,N), (1,1,N,N) in `dtype`.  state_with_delay.py:29-35"""
    values = np.asarray(values)
    network = np.asarray(network)
    n, f = values.shape
    assert network.shape == (n, n)                       # :25
    assert np.sum(np.diag(network)) == 0                 # :26
    v = values.transpose(1, 0).reshape(1, 1, f, n).astype(dtype)
    a = network.reshape(1, 1, n, n).astype(dtype)
    return v, a


def gso_powers(A, K, dtype=np.float32):
    """A: (B,N,N) -> curr_gso (B,K,N,N): I, A, A@A, ... (left-multiplied).  :38-41"""
    A = np.asarray(A, dtype=dtype)
    B, N, _ = A.shape
    out = np.zeros((B, K, N, N), dtype=dtype)
    out[:, 0] = np.eye(N, dtype=dtype)
    for j in range(1, K):
        out[:, j] = np.matmul(A, out[:, j - 1])
    return out


def gso_update(A, G_prev, X_t, Xd_prev, K, dtype=np.float32):
    """One state transition for B episodes.

    A      (B,N,N)     adjacency at time t (already cast)
    G_prev (B,K,N,N)   previous delay_gso, or None at episode start
    X_t    (B,F,N)     features at time t (already transposed)
    Xd_prev(B,K,F,N)   previous delay_state, or None
    returns delay_gso (B,K,N,N), delay_state (B,K,F,N)
    """
    A = np.asarray(A, dtype=dtype)
    X_t = np.asarray(X_t, dtype=dtype)
    B, N, _ = A.shape
    F = X_t.shape[1]
    G = np.zeros((B, K, N, N), dtype=dtype)
    G[:, 0] = np.eye(N, dtype=dtype)                                   # :45
    Xd = np.zeros((B, K, F, N), dtype=dtype)
    Xd[:, 0] = X_t                                                     # :51
    if G_prev is not None and K > 1:
        Gp = np.asarray(G_prev, dtype=dtype)
        G[:, 1:K] = np.matmul(A[:, None, :, :], Gp[:, 0:K - 1])        # :47
    if Xd_prev is not None and K > 1:
        Xd[:, 1:K] = np.asarray(Xd_prev, dtype=dtype)[:, 0:K - 1]      # :53
    return G, Xd
