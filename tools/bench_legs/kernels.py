"""bench.py leg: live HBM-roofline measurements of the dense-contract kernels (HIP events over rotating input sets)."""
import numpy as np
import torch

from multiagent_gnn_policies_amd import ops
from multiagent_gnn_policies_amd.envs import FlockParams
from multiagent_gnn_policies_amd.learner.state_with_delay import BatchedDelayState

from .common import F_FEAT, N_ACT


def time_kernel(fn, n_sets, iters):
    """Average duration (ms) of one launch of fn(i): HIP events on the launch stream around a captured HIP graph of
    `n_sets` back-to-back launches (one per rotating input set), replayed until `iters` launches have run.  The graph
    keeps the measurement GPU-bound (an eager Python loop is host-bound below ~25 us per launch); what remains on
    top of the kernel is the ~1.5 us dependent-kernel boundary, so the figure is slightly conservative."""
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for i in range(min(n_sets, 3)):
            fn(i)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with ops.graph_capture(graph):
        for i in range(n_sets):
            fn(i)
    reps = max(2, (iters + n_sets - 1) // n_sets)
    graph.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        graph.replay()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / (reps * n_sets)
    del graph
    return ms


def kernel_rooflines(device, B, N, K, actor, flock_c):
    """Per-kernel live measurements on rotating buffers (working set > 256 MiB Infinity Cache)."""
    g_bytes = 4 * K * N * N * B
    n_sets = max(2, int(np.ceil(320 * 2 ** 20 / g_bytes)))
    n_sets = min(n_sets, 24)
    gen = torch.Generator(device=device).manual_seed(1)
    Gs = [torch.rand((B, K, N, N), device=device, generator=gen) for _ in range(n_sets)]
    Xs = [torch.randn((B, K, F_FEAT, N), device=device, generator=gen) for _ in range(n_sets)]
    res = {}
    # --- aggregation (the roofline kernel)
    iters = max(50, 4 * n_sets)
    ms = time_kernel(lambda i: ops.agg_fwd(Xs[i].permute(0, 2, 1, 3), Gs[i]), n_sets, iters)
    agg_bytes = (4 * K * N * N + 8 * K * F_FEAT * N) * B
    res['agg_fwd'] = dict(ms=ms, bytes=agg_bytes, gbs=agg_bytes / ms / 1e6)
    # --- whole Actor forward (aggregation + filter GEMM + MLP readout), fused kernel
    with torch.no_grad():
        ms = time_kernel(lambda i: actor(Xs[i], Gs[i]), n_sets, iters)
    n_params = sum(p.numel() for p in actor.parameters())
    act_bytes = (4 * K * N * N + 4 * K * F_FEAT * N + 4 * N_ACT * N) * B + 4 * n_params
    res['actor_fwd'] = dict(ms=ms, bytes=act_bytes, gbs=act_bytes / ms / 1e6)
    # --- delayed-GSO update: read A, G_prev[1..K-2]; write K slices
    As = [torch.zeros((B, N, N), device=device) for _ in range(n_sets)]
    for a in As:
        mask = torch.rand((B, N, N), device=device, generator=gen) < (8.0 / N)
        a.copy_(mask.float() / mask.float().sum(-1, keepdim=True).clamp(min=1))
    Gn = [torch.empty((B, K, N, N), device=device) for _ in range(2)]
    Xn = torch.empty((B, K, F_FEAT, N), device=device)
    Xt = torch.randn((B, F_FEAT, N), device=device, generator=gen)
    ms = time_kernel(lambda i: ops.gso_update_into(As[i], Gs[i], Gn[i % 2], Xt, Xs[i], Xn, True), n_sets, iters)
    gso_bytes = 4 * (2 * K - 1) * N * N * B
    res['gso_update'] = dict(ms=ms, bytes=gso_bytes, gbs=gso_bytes / ms / 1e6)
    # --- sim step: writes the dense N x N network matrix
    xs = torch.randn((B, N, 4), device=device, dtype=torch.float64, generator=gen) * 3.0
    u = torch.zeros((B, N, 2), device=device)
    feat = torch.empty((B, F_FEAT, N), device=device)
    rew = torch.empty((B,), device=device, dtype=torch.float64)
    ms = time_kernel(lambda i: ops.flock_step(xs, u, flock_c, A=As[i], feat=feat, reward=rew), n_sets, iters)
    sim_bytes = (4 * N * N + 8 * 4 * N * 2 + 4 * (2 + 6) * N) * B
    res['flock_step'] = dict(ms=ms, bytes=sim_bytes, gbs=sim_bytes / ms / 1e6)
    # --- fused sim step + delayed-GSO / delay-line transition (what the timed step actually launches):
    #     writes A_t (N^2) and the products (K-2) N^2, gathers (K-2) N^2 of G_prev, features/labels/state are small
    try:
        from multiagent_gnn_policies_amd.envs import VecFlock
        params = FlockParams(n_agents=N, init_mode='grid')
        sim = VecFlock(B, params, device, with_expert=True)
        sim.reset(np.random.RandomState(3))
        states = [BatchedDelayState(device, B, K, F_FEAT, N) for _ in range(min(n_sets, 6))]
        for st_ in states:
            st_.push(sim.network, sim.features)
            sim.step_advance(u.view(B, N, 2), st_)
        ms = time_kernel(lambda i: sim.step_advance(u.view(B, N, 2), states[i % len(states)]), len(states), iters)
        ss_bytes = (4 * N * N * (1 + 2 * max(K - 2, 0)) + 8 * 4 * N * 2 + 4 * (2 + 6 + 2) * N + 4 * 2 * (K - 1) * F_FEAT * N) * B
        res['sim_state_step'] = dict(ms=ms, bytes=ss_bytes, gbs=ss_bytes / ms / 1e6)
    except Exception as e:                               # shapes the fused kernel does not cover
        res['sim_state_step'] = dict(ms=float('nan'), bytes=0, gbs=0.0, note=str(e))
    del Gs, Xs, As
    torch.cuda.empty_cache()
    return res, n_sets
