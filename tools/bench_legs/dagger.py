"""bench.py legs: --dagger-update (DAGGER updates / collection at the reference's training shape) and --dagger (BASELINE
configs[3]: one DAGGER round per rank, gradients exchanged between the ranks)."""
import gc
import os
import sys
import time

import numpy as np
import torch

from multiagent_gnn_policies_amd import ops, parallel
from multiagent_gnn_policies_amd.envs import FlockParams, VecFlock
from multiagent_gnn_policies_amd.learner.state_with_delay import BatchedDelayState

from .common import F_FEAT, N_ACT
from .launch import dist_record, emit_json


def dagger_update_bench():
    """Secondary measurement (`bench.py --dagger-update`): one DAGGER gradient_step at the reference's training
    shape (cfg/dagger.cfg: B=20, N=100, K=3) on the HIP path -- fused forward + MSE gradient + fused backward + flat
    Adam, replayed from one HIP graph -- next to the same op sequence of the CPU port (torch autograd + Adam; this is
    the cpu_baseline leg of the update measurement: the only place this function touches oracle/)."""
    import configparser
    from multiagent_gnn_policies_amd.learner.gnn_dagger import DAGGER
    B, N, K = 20, 100, 3
    cp = configparser.ConfigParser()
    cp['DEFAULT'] = dict(n_states='6', n_actions='2', k=str(K), hidden_size='32', gamma='0.99', tau='0.5',
                         n_agents=str(N), actor_lr='5e-5')
    cp['t'] = {}
    torch.manual_seed(11)
    dev = torch.device('cuda:0')
    learner = DAGGER(dev, cp['t'])
    gen = torch.Generator(device=dev).manual_seed(0)
    xd = torch.randn((B, K, F_FEAT, N), device=dev, generator=gen)
    mask = torch.rand((B, K, N, N), device=dev, generator=gen) < (8.0 / N)
    gd = mask.float() / mask.float().sum(-1, keepdim=True).clamp(min=1)
    gd[:, 0] = torch.eye(N, device=dev)
    yd = torch.randn((B, 1, N_ACT, N), device=dev, generator=gen)
    for _ in range(20):
        learner.gradient_step_tensors(xd, gd, yd)
    torch.cuda.synchronize()
    n = 300
    t0 = time.perf_counter()
    for _ in range(n):
        learner.gradient_step_tensors(xd, gd, yd)                 # drop-in semantics: the host reads every loss
    torch.cuda.synchronize()
    gpu_ms = 1e3 * (time.perf_counter() - t0) / n
    t0 = time.perf_counter()
    for _ in range(n):
        learner.gradient_step_tensors(xd, gd, yd, sync=False)     # vectorised DAGGER: losses stay on the device
    torch.cuda.synchronize()
    gpu_ms_pipe = 1e3 * (time.perf_counter() - t0) / n
    # vectorised DAGGER's round of updates: minibatches gathered from a device replay inside the kernel, index table
    # uploaded once, one graph replay per update (sampling on the host included: random.sample per update)
    from multiagent_gnn_policies_amd.learner.vec_dagger import DeviceReplay, IndexedUpdates
    cap, U = 4096, 2000
    rb = DeviceReplay(cap, K, F_FEAT, N, N_ACT, dev)
    for i0 in range(0, cap, B):
        rb.insert_batch(xd, gd, yd)
    iu = IndexedUpdates(learner, rb, B, U)
    iu.run_sampled(64)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    loss_round = iu.run_sampled(U).item()
    gpu_ms_idx = 1e3 * (time.perf_counter() - t0) / U
    assert np.isfinite(loss_round)
    from oracle import torch_port                          # CPU leg
    res = {}
    xc, gc, yc = xd.cpu(), gd.cpu(), yd.cpu()
    for thr in sorted({1, torch.get_num_threads()}):
        torch.set_num_threads(thr)
        Ws = [torch.nn.Parameter(c.weight.detach().cpu().clone()) for c in learner.actor.conv_layers]
        bs = [torch.nn.Parameter(c.bias.detach().cpu().clone()) for c in learner.actor.conv_layers]
        opt = torch.optim.Adam(Ws + bs, lr=5e-5)

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
    # DAGGER data collection (BASELINE.json configs[3], gnn_dagger.py:154-178): rollouts with expert labels, beta coin and
    # replay insert -- on the collecting build of the resident kernel (one launch per round) vs the host-stepped two-launch
    # loop of round 1 (>= 5 launches + host RNG + H2D per step)
    from multiagent_gnn_policies_amd.learner.vec_dagger import FrameReplay, collect_round, FrameUpdates
    lanes, Tc = 256, 500
    pcol = FlockParams(n_agents=N, init_mode='grid')
    simc = VecFlock(lanes, pcol, dev, with_expert=True)
    stc = BatchedDelayState(dev, lanes, K, F_FEAT, N)
    memc = FrameReplay(lanes, lanes * Tc, K, N, dev)
    beta_t = torch.full((lanes,), 0.75, device=dev)
    eps = torch.arange(lanes, dtype=torch.int32, device=dev)
    np.random.seed(3)
    collect_round(learner, simc, stc, memc, beta_t, eps, 11, 20)                 # warm-up (also the reset sampling cache)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    collect_round(learner, simc, stc, memc, beta_t, eps, 11, Tc)
    torch.cuda.synchronize()
    t_round = time.perf_counter() - t0
    e0c, e1c = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    from multiagent_gnn_policies_amd.learner.rollouts import _actor_params
    wsc, bsc = _actor_params(learner.actor)
    img = ops.rollout_image(wsc, bsc, tuple(learner.actor.layers), K, N)
    exp_io = simc.controller().permute(0, 2, 1).contiguous()
    e0c.record()
    ops.rollout_collect(simc.x, stc._G[stc._cur], stc.delay_state, tuple(learner.actor.layers), simc._c, Tc, memc, exp_io, beta_t,
                        eps, 11, age0=Tc, ring_step0=memc.head, carry=stc.carry_buffer(),
                        flags=ops.RO_ENTER_CARRY | ops.RO_EXIT_CARRY | ops.RO_SKIP_DENSE, image=img)
    e1c.record()
    torch.cuda.synchronize()
    collect_kernel_ms = e0c.elapsed_time(e1c)
    fu = FrameUpdates(learner, memc, B, 2000, True)
    fu.run_sampled(64)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    fu.run_sampled(2000).item()                                # random.sample per update, overlapped with the GPU's replays
    frame_update_ms = 1e3 * (time.perf_counter() - t0) / 2000

    return {"update": "DAGGER gradient_step B=20 N=100 K=3", "hip_ms": gpu_ms, "hip_updates_per_s": 1e3 / gpu_ms,
            "collect": {"lanes": lanes, "steps": Tc, "kernel_ms": collect_kernel_ms,
                        "kernel_agent_steps_per_s": lanes * N * Tc / (1e-3 * collect_kernel_ms),
                        "round_wall_s_incl_host_reset_sampling": t_round,
                        "replay_bytes_per_transition": memc.bytes_per_transition(),
                        "hip_ms_frame_update_round": frame_update_ms},
            "hip_ms_pipelined": gpu_ms_pipe, "hip_updates_per_s_pipelined": 1e3 / gpu_ms_pipe,
            "hip_ms_indexed_round": gpu_ms_idx, "hip_updates_per_s_indexed_round": 1e3 / gpu_ms_idx,
            "cpu_port_ms_by_threads": res, "host_cores": os.cpu_count()}


def dagger_round_bench(args, device, rank, world):
    """BASELINE.json configs[3] (reference gnn_dagger.py:126-243, one device, one env): one DAGGER round on every rank --
      collection  --episodes lanes x --steps env steps inside ONE mgp_rollout_collect launch per rank (policy forward, expert
                  label, beta coin, simulator step, state transition, frame filed into the replay ring); ranks never talk
      updates     --updates minibatch updates of --batch-size samples PER RANK from the rank's own replay, captured 32 to a
                  HIP graph; the ranks' gradients (1,730 floats + the loss) are exchanged inside every update -- the one-shot
                  IPC exchange (csrc/p2p_device.h) or, without it, the RCCL all-reduce captured in the graph
    Timed with the contract's barrier + synchronize bracketing, MAX over ranks; the weights must be bit-identical on every
    rank at the end (the run fails otherwise)."""
    import configparser
    from multiagent_gnn_policies_amd.learner.gnn_dagger import DAGGER
    from multiagent_gnn_policies_amd.learner.vec_dagger import (FrameReplay, FrameUpdates, collect_round, collect_supported,
                                                                _dp_mode)
    dist = torch.distributed
    lanes, N, K, T, U, Bt = args.episodes, args.agents, args.taps, args.steps, args.updates, args.batch_size
    cp = configparser.ConfigParser()
    cp['DEFAULT'] = dict(n_states=str(F_FEAT), n_actions=str(N_ACT), k=str(K), hidden_size=str(args.hidden),
                         n_layers=str(args.layers), gamma='0.99', tau='0.5', n_agents=str(N), actor_lr='5e-5')
    cp['t'] = {}
    torch.manual_seed(11)
    learner = DAGGER(device, cp['t'])
    if not (collect_supported(learner, K, N) and FrameUpdates.supported(learner, Bt, N)):
        raise SystemExit("bench.py --dagger: shape outside mgp_rollout_collect / the graph-captured update path")
    p = FlockParams(n_agents=N, init_mode=args.init)
    sim = VecFlock(lanes, p, device, with_expert=True)
    state = BatchedDelayState(device, lanes, K, F_FEAT, N)
    memory = FrameReplay(lanes, lanes * max(T, args.warmup, 1), K, N, device)
    beta = torch.full((lanes,), 0.75, device=device)
    eps = torch.arange(rank * lanes, (rank + 1) * lanes, dtype=torch.int32, device=device)
    np.random.seed(1000 + rank)
    import random
    random.seed(1000 + rank)

    def barrier():
        if world > 1:
            dist.barrier()

    def max_over_ranks(x):
        if world > 1:
            t = torch.tensor([x], device=device, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            return float(t.item())
        return x

    # ---- collection: reset sampling (host, once per round) is outside the timed region, the launch inside
    from multiagent_gnn_policies_amd.learner.rollouts import _actor_params

    def collect_factored(steps):
        """N > 256: the same round on the factored state in HBM (K launches per env step enqueued by one library call;
        frame, label and coin inside the policy launch: mgp_sparse_policy_collect)."""
        from multiagent_gnn_policies_amd.learner.sparse_rollout import SparseFlockState, sparse_collect
        sim.reset(np.random)
        state.reset()
        state.push(sim.network, sim.features)
        sp = SparseFlockState(sim, K)
        sp.observe_reset(sim)
        gc.disable()                                             # (see timed(): no interpreter GC pass inside the timed region)
        barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        sparse_collect(learner.actor, sim, sp, memory, beta, eps, 11, 0, steps)
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        gc.enable()
        barrier()
        return max_over_ranks(el)

    def collect(steps):
        if N > 256:
            return collect_factored(steps)
        sim.reset(np.random)
        state.reset()
        state.push(sim.network, sim.features)
        expert_io = sim.controller().permute(0, 2, 1).contiguous()
        ws, bs = _actor_params(learner.actor)
        image = ops.rollout_image(ws, bs, tuple(learner.actor.layers), K, N)
        carry = state.carry_buffer()
        gc.disable()
        barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ok = ops.rollout_collect(sim.x, state._G[state._cur], state.delay_state, tuple(learner.actor.layers), sim._c, steps,
                                 memory, expert_io, beta, eps, 11, age0=0, ring_step0=memory.head, carry=carry,
                                 flags=ops.RO_ENTER_CARRY | ops.RO_EXIT_CARRY | ops.RO_SKIP_DENSE, image=image)
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        gc.enable()
        barrier()
        assert ok
        memory.advance(steps)
        state._pushes += steps
        state._dense_stale = True
        return max_over_ranks(el)
    gc.freeze()
    collect(max(args.warmup, K))
    t_collect = collect(T)
    # ---- updates
    fu = FrameUpdates(learner, memory, Bt, max(U, 64), p.mean_pooling)
    learner.begin_updates()
    gc.freeze()
    fu.run_sampled(64)                                           # warm-up: captures both graphs
    learner.end_updates()
    gc.disable()
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    loss_sum = fu.run_sampled(U)
    torch.cuda.synchronize()
    t_upd = time.perf_counter() - t0
    gc.enable()
    barrier()
    t_upd = max_over_ranks(t_upd)
    learner.end_updates()
    loss_mean = float(loss_sum.item()) / U
    # ---- every rank must hold the same weights, bit for bit
    identical = True
    if world > 1:
        cdev = device if dist.get_backend() == 'nccl' else torch.device('cpu')
        mine = learner.actor_optim.flat.detach().to(cdev)
        parts = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(parts, mine)
        identical = all(torch.equal(parts[0], q) for q in parts[1:])
    if rank == 0:
        out = {
            "metric": "agent-steps/sec of DAGGER data collection, FlockingRelative-v0 N=%d K=%d" % (N, K),
            "value": world * lanes * N * T / t_collect, "unit": "agent-steps/s", "n_gpus": world, "steps": T,
            "warmup": args.warmup, "ms_per_step": 1e3 * t_collect / T, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "DAGGER round (BASELINE.json configs[3]): %d lanes x %d steps of data collection per rank "
                                   "(%s: policy forward, expert label, beta coin, sim step, frame insert), "
                                   "then %d updates of %d samples per rank with the gradient exchanged between %d rank(s)"
                                   % (lanes, T, "mgp_rollout_collect" if N <= 256 else "factored state, mgp_sparse_policy_collect",
                                      U, Bt, world),
                       "episodes_per_gpu": lanes, "agents": N, "taps": K, "hidden": [args.hidden] * args.layers,
                       "init": args.init, "beta": 0.75,
                       "parallelism": "episodes sharded x%d; one exchange of %d floats per update"
                                      % (world, learner.actor_optim.flat.numel() + 1)},
            "updates": {"count": U, "batch_size_per_rank": Bt, "ms_per_update": 1e3 * t_upd / U,
                        "updates_per_s": U / t_upd, "samples_per_s": U * Bt * world / t_upd, "mean_loss": loss_mean,
                        "exchange": (fu.dp or "none (single process)"),
                        "exchange_mem_kind": getattr(learner.p2p, 'mem_kind', None),
                        "exchange_bringup": parallel.P2PExchange.last_bringup,
                        "updates_per_graph": 32,
                        # aggregated: mgp_replay_aggregate + mgp_train_step_agg (the K-hop products along the frames' bit rows,
                        # operator slices never formed); dense: mgp_replay_gather_many / _rows + mgp_train_step_indexed
                        "slots": "aggregated" if fu.aggregated else "dense"},
            "round_s": t_collect + t_upd,
            "weights_bit_identical_across_ranks": identical,
            "dist": dist_record(),
        }
        emit_json(out)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if not identical:
        sys.stderr.write("bench.py --dagger: the ranks' weights differ\n")
        sys.exit(4)
