#!/usr/bin/env python3
"""Workload for rocprofv3 --pmc passes: every hot kernel of the step, 20 launches each, on rotating input sets
larger than the Infinity Cache (same shapes as bench.py's roofline leg).  Usage on the GPU box:
    rocprofv3 --pmc FETCH_SIZE --kernel-trace -d out -o pmc_fetch -- python tools/pmc_probe.py
    rocprofv3 --pmc WRITE_SIZE --kernel-trace -d out -o pmc_write -- python tools/pmc_probe.py
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from multiagent_gnn_policies_amd.envs import FlockParams  # noqa: E402
from multiagent_gnn_policies_amd.learner import Actor  # noqa: E402


def main():
    B = int(os.environ.get('PROBE_B', '256'))
    N = int(os.environ.get('PROBE_N', '100'))
    K = 3
    dev = torch.device('cuda:0')
    if os.environ.get('PROBE_FACTORED'):
        # the factored path (N > 256): simulator + gather + policy launches of 200 env steps at BASELINE configs[2]'s shape
        ro = bench.Rollout(dev, B, N, K, [32, 32], seed=1000)
        assert ro.factored_supported()
        # [r6] six calls of PROBE_FT steps each (the first enters from the reset observation): where the persistent form covers
        # the shape every call is ONE launch of spp_rollout_kernel, whose bytes per launch / PROBE_FT are the bytes per env step
        FT = int(os.environ.get('PROBE_FT', '200'))
        for _ in range(6):
            ro.run_resident(FT)
        torch.cuda.synchronize()
        print('factored path: 6 calls of', FT, 'env steps at', B, 'x', N)
        return
    actor = Actor(6, 2, [32, 32], K, 0).to(dev)
    bench.load_weights(actor)
    actor.eval()
    if not os.environ.get('PROBE_ROLLOUT_ONLY'):
        res, n_sets = bench.kernel_rooflines(dev, B, N, K, actor, FlockParams(n_agents=N).to_c())
        print({k: round(v['ms'] * 1e3, 2) for k, v in res.items()}, n_sets)
    # episode-resident rollout kernel: a few launches of PROBE_T steps each (bench.py's default timed launch)
    T = int(os.environ.get('PROBE_T', '1000'))
    ro = bench.Rollout(dev, B, N, K, [32, 32], seed=1000)
    if ro.resident_supported():
        ro.run_resident(3)                 # the first launch enters from the reset observation; the profiled ones carry on
        for _ in range(5):
            ro.run_resident(T)
        torch.cuda.synchronize()
        print('rollout_kernel: 5 launches of', T, 'steps')


if __name__ == '__main__':
    main()
