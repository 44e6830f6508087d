#!/usr/bin/env python3
"""DAGGER update throughput (reference learner/gnn_dagger.py:76-96 at cfg/dagger.cfg: B=20, N=100, K=3):
HIP path (fused forward + backward + MSE + flat Adam) vs the torch-CPU reference op sequence on this host."""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import configparser
    from multiagent_gnn_policies_amd.learner.gnn_dagger import DAGGER
    from oracle import synth, torch_port
    B, N, K = 20, 100, 3
    cp = configparser.ConfigParser()
    cp['DEFAULT'] = dict(n_states='6', n_actions='2', k=str(K), hidden_size='32', gamma='0.99', tau='0.5',
                         n_agents=str(N), actor_lr='5e-5')
    cp['t'] = {}
    torch.manual_seed(11)
    learner = DAGGER(torch.device('cuda:0'), cp['t'])
    X, G = synth.make_inputs(0, B, K, 6, N)
    Y = np.random.RandomState(1).randn(B, 1, 2, N).astype(np.float32)
    xd, gd, yd = (torch.from_numpy(a).cuda() for a in (X, G, Y))
    for _ in range(20):
        learner.gradient_step_tensors(xd, gd, yd)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 300
    for _ in range(n):
        learner.gradient_step_tensors(xd, gd, yd)
    torch.cuda.synchronize()
    gpu_ms = 1e3 * (time.perf_counter() - t0) / n
    # CPU reference op sequence (torch autograd + Adam), 1 thread and all threads
    res = {}
    for thr in sorted({1, torch.get_num_threads()}):
        torch.set_num_threads(thr)
        Ws = [torch.nn.Parameter(c.weight.detach().cpu().clone()) for c in learner.actor.conv_layers]
        bs = [torch.nn.Parameter(c.bias.detach().cpu().clone()) for c in learner.actor.conv_layers]
        opt = torch.optim.Adam(Ws + bs, lr=5e-5)
        xc, gc, yc = torch.from_numpy(X), torch.from_numpy(G), torch.from_numpy(Y)

        def step():
            opt.zero_grad()
            out = torch_port.actor_forward(xc, gc, Ws, bs, 0, K)
            loss = torch.nn.functional.mse_loss(out, yc)
            loss.backward()
            opt.step()
            return loss.item()
        for _ in range(5):
            step()
        t0 = time.perf_counter()
        m = 100
        for _ in range(m):
            step()
        res[thr] = 1e3 * (time.perf_counter() - t0) / m
    print(json.dumps({"update": "DAGGER gradient_step B=20 N=100 K=3", "hip_ms": gpu_ms, "hip_updates_per_s": 1e3 / gpu_ms,
                      "cpu_port_ms_by_threads": res, "host_cores": os.cpu_count()}))


if __name__ == '__main__':
    main()
