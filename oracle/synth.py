"""Seeded synthetic (S, X) inputs shared by goldens, tests and bench (TEST INFRASTRUCTURE).

SURVEY.md section 8(d): X ~ N(0,1) fp32; G per (b,j) from random geometric graphs with
comm radius R = 1, target mean degree ~8, zero diagonal, row-normalised by max(deg,1)
(mean pooling); G_0 = I, G_j = A^(j) @ G_{j-1} with an independent graph per factor.
Everything derives from numpy.random.RandomState(seed) so goldens only need to store
seeds + outputs.
"""
import numpy as np


def geometric_adjacency(rs, n, mean_degree=8.0, radius=1.0):
    """One row-normalised radius graph (n,n) fp64 with zero diagonal."""
    area = n * np.pi * radius * radius / mean_degree
    rad = np.sqrt(area / np.pi)
    length = rad * np.sqrt(rs.uniform(0.0, 1.0, size=n))
    angle = rs.uniform(0.0, 2.0 * np.pi, size=n)
    px, py = length * np.cos(angle), length * np.sin(angle)
    dx = px[:, None] - px[None, :]
    dy = py[:, None] - py[None, :]
    r2 = dx * dx + dy * dy
    np.fill_diagonal(r2, np.inf)
    adj = (r2 < radius * radius).astype(np.float64)
    deg = adj.sum(axis=1)
    deg[deg == 0] = 1.0
    return adj / deg[:, None]


def make_adjacency_batch(seed, B, N, mean_degree=8.0):
    """(B,N,N) fp32 adjacencies."""
    rs = np.random.RandomState(seed)
    return np.stack([geometric_adjacency(rs, N, mean_degree) for _ in range(B)]).astype(np.float32)


def make_inputs(seed, B, K, F, N, mean_degree=8.0):
    """Returns X (B,K,F,N) fp32 and G (B,K,N,N) fp32 (delayed products of independent graphs)."""
    rs = np.random.RandomState(seed)
    X = rs.randn(B, K, F, N).astype(np.float32)
    G = np.zeros((B, K, N, N), dtype=np.float32)
    for b in range(B):
        G[b, 0] = np.eye(N, dtype=np.float32)
        for j in range(1, K):
            A = geometric_adjacency(rs, N, mean_degree).astype(np.float32)
            G[b, j] = A @ G[b, j - 1]
    return X, G


def make_dense_inputs(seed, B, K, F, N):
    """Fully dense random G (no structure) -- exercises the kernels without zero-skipping luck."""
    rs = np.random.RandomState(seed)
    X = rs.randn(B, K, F, N).astype(np.float32)
    G = (rs.randn(B, K, N, N) / np.sqrt(N)).astype(np.float32)
    return X, G


def make_weights(seed, n_s, n_a, hidden_layers, k, ind_agg, scale=None):
    """Deterministic weights in the reference state_dict layout (NOT torch's default init;
    used where bit-identical torch RNG consumption does not matter)."""
    rs = np.random.RandomState(seed + 7919)
    layers = [n_s] + list(hidden_layers) + [n_a]
    Ws, bs = [], []
    for i in range(len(layers) - 1):
        step = k if i == ind_agg else 1
        fan_in = layers[i] * step
        s = scale if scale is not None else 1.0 / np.sqrt(fan_in)
        Ws.append(rs.uniform(-s, s, size=(layers[i + 1], layers[i], step, 1)).astype(np.float32))
        bs.append(rs.uniform(-s, s, size=(layers[i + 1],)).astype(np.float32))
    return Ws, bs


def checksum(*arrays):
    """Order-sensitive fp64 checksum used to detect generator drift against goldens."""
    tot = 0.0
    for a in arrays:
        a = np.asarray(a, dtype=np.float64).ravel()
        w = np.cos(np.arange(a.size, dtype=np.float64) * 0.37) + 1.5
        tot += float(np.dot(a, w))
    return tot
