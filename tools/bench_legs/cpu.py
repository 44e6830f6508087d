"""bench.py leg: cpu_baseline -- the reference-style single-episode loop of the CPU port, bounded sample."""
import os
import time

import numpy as np
import torch

from .common import F_FEAT, N_ACT, ROOT


def cpu_baseline(N, K, hidden, budget_s=12.0, init_mode='auto', actor=None, variant=None):
    """Reference-style single-episode CPU loop (oracle/torch_port.py + numpy sim), bounded sample.  `actor`: the policy the
    GPU legs ran (its weights are copied to the host); `variant`: FlockParams fields of the environment variant."""
    from oracle import flock as ofl, torch_port
    path = os.path.join(ROOT, 'tests', 'golden', 'ckpt_dagger_k3.npz')
    torch.manual_seed(11)
    if actor is not None:
        Ws = [c.weight.detach().cpu().clone() for c in actor.conv_layers]
        bs = [c.bias.detach().cpu().clone() for c in actor.conv_layers]
    elif os.path.exists(path) and K == 3 and hidden == [32, 32]:
        with np.load(path) as z:
            Ws = [torch.from_numpy(z[f'conv_layers__{i}__weight']) for i in range(3)]
            bs = [torch.from_numpy(z[f'conv_layers__{i}__bias']) for i in range(3)]
    else:
        dims = [F_FEAT] + hidden + [N_ACT]
        Ws = [torch.randn(dims[i + 1], dims[i], K if i == 0 else 1, 1) * 0.1 for i in range(len(dims) - 1)]
        bs = [torch.zeros(dims[i + 1]) for i in range(len(dims) - 1)]
    p = ofl.FlockParams(n_agents=N, init_mode=init_mode, **(variant or {}))
    x0 = ofl.reset(np.random.RandomState(0), p)
    default_threads = torch.get_num_threads()
    runs = []
    for threads in sorted({1, default_threads}):
        torch.set_num_threads(threads)
        x = x0
        torch_port.rollout_steps(x, p, Ws, bs, K, 5)                     # warm-up
        chunk, done = 50, 0
        t0 = time.perf_counter()
        while True:
            _, x = torch_port.rollout_steps(x, p, Ws, bs, K, chunk)
            done += chunk
            el = time.perf_counter() - t0
            if el >= budget_s / 2 or done >= 20000:
                break
        runs.append((N * done / el, threads, done, el))
    torch.set_num_threads(default_threads)
    best = max(runs)
    return dict(value=best[0], unit='agent-steps/s', cores=best[1], kind='port',
                sample='1 episode x %d steps (%.1f s) at %d torch thread(s), reference-style B=1 loop: numpy fp64 '
                       'sim + torch-CPU state update (incl. curr_gso) + Actor forward; all thread settings tried: '
                       '%s; host has %d logical cores'
                       % (best[2], best[3], best[1],
                          ', '.join('%d thr -> %.3g agent-steps/s' % (r[1], r[0]) for r in runs), os.cpu_count() or 0),
                ms_per_env_step=1e3 * best[3] / best[2])
