#!/usr/bin/env python3
"""Dense-contract aggregation (mgp_agg_fwd) and fused Actor forward (mgp_actor_fwd) at several batch sizes: achieved HBM rate.
    MGP_AGG_FORM=8 python tools/gpu/agg_ab.py     # the eight-wave kernel (round 2/3)
    MGP_AGG_FORM=4 python tools/gpu/agg_ab.py     # four waves per (episode, tap)
Rotating input sets of > 256 MiB in total (no L2 / MALL reuse between launches), HIP events around a graph of back-to-back launches."""
import os
import sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from multiagent_gnn_policies_amd import ops  # noqa: E402

dev = torch.device('cuda:0')
for B, N, K in [tuple(int(v) for v in t.split(',')) for t in os.environ.get('AGG_SHAPES', '256,100,3 512,100,3 1024,100,3 2048,100,3 256,64,3 256,128,3 256,200,4 64,1000,3').split()]:
    F = 6
    per = 4 * B * K * N * N
    n_sets = max(2, min(12, (320 << 20) // per + 1))
    Gs = [torch.rand((B, K, N, N), device=dev) for _ in range(n_sets)]
    Xs = [torch.randn((B, K, F, N), device=dev) for _ in range(n_sets)]
    Ys = [torch.empty((B, K, F, N), device=dev) for _ in range(n_sets)]

    def fn(i):
        L = ops._lib.lib()
        X, G, Y = Xs[i], Gs[i], Ys[i]
        ops._lib.check(L.mgp_agg_fwd(ops._ptr(X), ops._ptr(G), ops._ptr(Y), B, K, F, N, K * F * N, F * N, N, K * F * N, F * N, N,
                                     ops._stream()), 'agg')
    ms = bench.time_kernel(fn, n_sets, max(40, 4 * n_sets))
    by = (4 * K * N * N + 8 * K * F * N) * B
    print('MGP_AGG_FORM=%s agg_fwd B=%d N=%d K=%d: %.2f us  %.2f TB/s  frac %.3f' % (os.environ.get('MGP_AGG_FORM', '4'), B, N, K, 1e3 * ms,
                                                                                 by / ms / 1e9, by / ms / 1e9 / 8.0), flush=True)
    if N <= 128 and K == 3:                                      # the fused Actor forward on the shipped checkpoint's shape
        from multiagent_gnn_policies_amd.learner import Actor
        actor = Actor(6, 2, [32, 32], K, 0).to(dev)
        bench.load_weights(actor)
        actor.eval()

        def fa(i):
            with torch.no_grad():
                actor(Xs[i], Gs[i])
        fa(0)
        ms = bench.time_kernel(fa, n_sets, max(40, 4 * n_sets))
        by = (4 * K * N * N + 4 * K * F * N + 4 * 2 * N) * B
        print('MGP_ACTOR_POL=%s actor_fwd B=%d N=%d K=%d: %.2f us  %.2f TB/s  frac %.3f' % (os.environ.get('MGP_ACTOR_POL', '1'), B, N, K, 1e3 * ms,
                                                                                      by / ms / 1e9, by / ms / 1e9 / 8.0), flush=True)
    del Gs, Xs, Ys
    torch.cuda.empty_cache()
