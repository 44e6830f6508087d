#!/usr/bin/env python3
"""Where the wall time of bench.py's 20-step timed region goes: plain launch / kernel-stamped events / recorded events."""
import os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
from multiagent_gnn_policies_amd import _lib

ro = bench.Rollout(torch.device('cuda:0'), 256, 100, 3, [32, 32], seed=1000, init_mode='grid')
ro.prepare_resident([5, 20])
ro.run_resident(5)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); e1.record(); torch.cuda.synchronize()
L = _lib.lib()
res = {m: [] for m in ('none', 'ext', 'torch')}
for it in range(20):
    for mode in res:
        torch.cuda.synchronize()
        time.sleep(0.002)
        t0 = time.perf_counter()
        if mode == 'torch':
            e0.record()
        if mode == 'ext':
            L.mgp_set_launch_events(e0.cuda_event, e1.cuda_event)
        ro.run_resident(20)
        if mode == 'torch':
            e1.record()
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        gpu = e0.elapsed_time(e1) * 1e3 if mode != 'none' else float('nan')
        res[mode].append((1e6 * (t1 - t0), 1e6 * (t2 - t1), 1e6 * (t2 - t0), gpu))
for mode, rows in res.items():
    m = np.median(np.array(rows[3:]), axis=0)
    print('%-6s host enqueue %.1f us | wait %.1f us | total %.1f us | GPU (events) %.1f us' % (mode, *m))
