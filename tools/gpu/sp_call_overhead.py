#!/usr/bin/env python3
"""Per-call cost of the factored rollout from Python: t(T) = a + b T over calls of T = 20, 100, 500 steps (64 x 1000, K = 3)."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
dev = torch.device('cuda:0')
ro = bench.Rollout(dev, 64, 1000, 3, [32, 32], seed=1000)
ro.run_resident(20)
torch.cuda.synchronize()
for T in (20, 100, 500, 100, 20):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    e0.record()
    for _ in range(5):
        ro.run_resident(T)
    e1.record()
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    print('T=%d: %.1f us per call (events), %.1f us per call (host wall), %.2f us per step' % (T, 1e3 * e0.elapsed_time(e1) / 5, 1e6 * (t1 - t0) / 5,
                                                                                         1e3 * e0.elapsed_time(e1) / 5 / T), flush=True)
