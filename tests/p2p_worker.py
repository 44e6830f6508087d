"""Worker of tests/test_gpu_p2p.py: one rank of a W-rank run of the one-shot gradient exchange (mgp_p2p_*), ranks
sharing whatever devices are visible (LOCAL_RANK folded onto them; on the one-GPU test box both ranks use cuda:0 --
hipIpc works between processes on one device, RCCL does not).  The process group (gloo) only carries the IPC handles and
the reference values.  Usage: RANK/WORLD_SIZE/MASTER_* in the env;  python p2p_worker.py <scenario>"""
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from multiagent_gnn_policies_amd import parallel  # noqa: E402


def gather_cpu(t):
    parts = [torch.zeros_like(t.cpu()) for _ in range(dist.get_world_size())]
    dist.all_gather(parts, t.cpu())
    return parts


def ordered_mean(parts):
    s = torch.zeros_like(parts[0])
    for p in parts:                       # rank order, fp32, then ONE division: the kernel's arithmetic
        s = s + p
    return s / float(len(parts))


def scenario_allreduce(comm, dev, rk, world):
    n = comm.n_floats
    gen = torch.Generator(device='cpu').manual_seed(100 + rk)
    # (i) many back-to-back exchanges, exact against the rank-ordered sum; odd sizes too
    for it in range(40):
        m = n if it % 3 else max(1, n - 7 * it)
        mine = torch.randn((m,), generator=gen).to(dev) * (10.0 ** (it % 5 - 2))
        ref = ordered_mean(gather_cpu(mine))
        out = mine.clone()
        comm.allreduce_mean_(out)
        assert torch.equal(out.cpu(), ref), (it, float((out.cpu() - ref).abs().max()))
    # (ii) uneven load: one rank is late by tens of milliseconds, alternating -- the early rank's kernel polls meanwhile
    for it in range(6):
        mine = torch.randn((n,), generator=gen).to(dev)
        ref = ordered_mean(gather_cpu(mine))
        if it % world == rk:
            time.sleep(0.05)
        out = mine.clone()
        comm.allreduce_mean_(out)
        torch.cuda.synchronize()
        assert torch.equal(out.cpu(), ref), it
    # (iii) 32 exchanges captured in one HIP graph, replayed 3 times (the sequence number lives on the device)
    bufs = [torch.randn((n,), generator=gen).to(dev) for _ in range(32)]
    refs = [ordered_mean(gather_cpu(b)) for b in bufs]
    work = [b.clone() for b in bufs]
    torch.cuda.synchronize()
    dist.barrier()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        for w in work:
            comm.allreduce_mean_(w)
    for rep in range(3):
        for w, b in zip(work, bufs):
            w.copy_(b)
        torch.cuda.synchronize()
        dist.barrier()
        graph.replay()
        torch.cuda.synchronize()
        for j, (w, r) in enumerate(zip(work, refs)):
            assert torch.equal(w.cpu(), r), (rep, j)
    # latency of one exchange inside the graph (all ranks in lock step)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    dist.barrier()
    e0.record()
    for _ in range(10):
        graph.replay()
    e1.record()
    torch.cuda.synchronize()
    us = 1e3 * e0.elapsed_time(e1) / 320
    st, seq = comm.status()
    assert st == 0
    return {"exchange_us_in_graph": us, "exchanges": seq, "mem_kind": comm.mem_kind}


def scenario_timeout(comm, dev, rk, world):
    """A peer that never shows up is an error status, not a hang."""
    from multiagent_gnn_policies_amd import _lib
    _lib.lib().mgp_p2p_set_timeout_ms(comm.handle, 100)
    res = {}
    if rk == 0:
        buf = torch.ones((comm.n_floats,), device=dev)
        t0 = time.perf_counter()
        comm.allreduce_mean_(buf)
        st, _ = comm.status()
        res = {"status": st, "wall_s": time.perf_counter() - t0}
        assert st == 1 and res["wall_s"] < 5.0, res
    dist.barrier()
    return res


def scenario_train(comm, dev, rk, world):
    """Data-parallel update with the exchange inside its second launch (mgp_train_step_p2p) against the same update
    computed by ONE process on the concatenated minibatch semantics: mean of the per-rank gradients, Adam."""
    import configparser
    from multiagent_gnn_policies_amd.learner.gnn_dagger import DAGGER
    cp = configparser.ConfigParser()
    cp['DEFAULT'] = dict(n_states='6', n_actions='2', k='3', hidden_size='32', gamma='0.99', tau='0.5', n_agents='100',
                         actor_lr='1e-3')
    cp['t'] = {}
    B, N, K = 20, 100, 3
    torch.manual_seed(5)
    learner = DAGGER(dev, cp['t'])
    assert learner.p2p is not None, "DAGGER must have brought the one-shot exchange up"
    data = []
    for q in range(world):                                        # every rank can build every rank's minibatch
        gen = torch.Generator(device='cpu').manual_seed(50 + q)
        X = torch.randn((B, K, 6, N), generator=gen)
        m = torch.rand((B, K, N, N), generator=gen) < 0.08
        G = m.float() / m.float().sum(-1, keepdim=True).clamp(min=1)
        G[:, 0] = torch.eye(N)
        Y = torch.randn((B, 1, 2, N), generator=gen)
        data.append((X.to(dev), G.to(dev), Y.to(dev)))
    # reference: local gradients of every rank through the same kernels (mgp_train_grads), mean in rank order, Adam
    parallel_is = parallel.is_distributed
    parallel.is_distributed = lambda: False
    torch.manual_seed(5)
    ref = DAGGER(dev, cp['t'])
    parallel.is_distributed = parallel_is
    ref.actor_optim.flat.copy_(learner.actor_optim.flat)
    losses_ref, losses = [], []
    for step in range(4):
        gs, ls = [], []
        for q in range(world):
            loss = ref._train_grads(*data[q])
            gs.append(ref.actor_optim.flat_grad.clone())
            ls.append(loss.clone())
        ref.actor_optim.flat_grad.copy_(ordered_mean(gs))
        ref.actor_optim.step()
        losses_ref.append(float(ordered_mean(ls).item()))
        losses.append(learner.gradient_step_tensors(*data[rk]))
    gu = learner._graphed[B]
    assert gu.p2p is not None and gu.graph is not None, "the data-parallel update must be one graph replay with the exchange inside"
    w = learner.actor_optim.flat.clone()
    assert int(learner.actor_optim.step_dev.item()) == 4
    assert np.allclose(losses, losses_ref, rtol=0, atol=1e-6), (losses, losses_ref)
    err = float((w - ref.actor_optim.flat).abs().max())
    assert err <= 1e-7, err
    parts = gather_cpu(w)
    assert all(torch.equal(parts[0], p) for p in parts[1:]), "weights must be bit-identical on every rank"
    learner.p2p.check()
    return {"losses": losses, "max_weight_diff_vs_single_process": err}


def scenario_dp_timeout(comm, dev, rk, world):
    """One rank falls behind INSIDE a data-parallel round (begin_updates .. end_updates) by more than the exchange's timeout (400 ms here).
    The early rank's poll gives up (Adam skipped for what it missed), the late rank finds every packet waiting and steps:
    only one of them sees an error locally.  end_updates() must raise on EVERY rank, with weights, both Adam moments and the
    step counter (device and host) back at the round's start and identical across ranks; after reset_exchange() the next
    round must work."""
    import configparser
    from multiagent_gnn_policies_amd import _lib
    from multiagent_gnn_policies_amd.learner.gnn_dagger import DAGGER
    cp = configparser.ConfigParser()
    cp['DEFAULT'] = dict(n_states='6', n_actions='2', k='3', hidden_size='32', gamma='0.99', tau='0.5', n_agents='100',
                         actor_lr='1e-3')
    cp['t'] = {}
    B, N, K = 20, 100, 3
    torch.manual_seed(5)
    learner = DAGGER(dev, cp['t'])
    assert learner.p2p is not None, "DAGGER must have brought the one-shot exchange up"
    gen = torch.Generator(device='cpu').manual_seed(50 + rk)
    X = torch.randn((B, K, 6, N), generator=gen).to(dev)
    m = torch.rand((B, K, N, N), generator=gen) < 0.08
    G = (m.float() / m.float().sum(-1, keepdim=True).clamp(min=1))
    G[:, 0] = torch.eye(N)
    G = G.to(dev)
    Y = torch.randn((B, 1, 2, N), generator=gen).to(dev)
    o = learner.actor_optim
    # the exchange's timeout is a kernel argument: it must be set BEFORE the update graph is captured (first update below)
    _lib.lib().mgp_p2p_set_timeout_ms(learner.p2p.handle, 400)
    # a good round first (captures the update graph)
    learner.begin_updates()
    for _ in range(2):
        learner.gradient_step_tensors(X, G, Y, sync=False)
    learner.end_updates()
    start = (o.flat.clone(), o.m.clone(), o.v.clone(), int(o.step_dev.item()), o.step_count)
    assert start[3] == 2 and start[4] == 2
    learner.begin_updates()
    learner.gradient_step_tensors(X, G, Y, sync=False)
    torch.cuda.synchronize()
    if rk == world - 1:
        time.sleep(2.0)                                          # the last rank is late: its peers' polls give up meanwhile
    for _ in range(2):
        learner.gradient_step_tensors(X, G, Y, sync=False)
    torch.cuda.synchronize()
    local_status = learner.p2p.status()[0]
    moved = not torch.equal(o.flat, start[0])                    # (some rank stepped: the round did run)
    raised = False
    try:
        learner.end_updates()
    except _lib.MgpError as e:
        raised = True
        msg = str(e)
    assert raised, "end_updates() must raise on every rank (local status %d)" % local_status
    assert torch.equal(o.flat, start[0]) and torch.equal(o.m, start[1]) and torch.equal(o.v, start[2])
    assert int(o.step_dev.item()) == start[3] and o.step_count == start[4]
    parts = gather_cpu(o.flat)
    assert all(torch.equal(parts[0], q) for q in parts[1:]), "rolled-back weights must be identical on every rank"
    seen = gather_cpu(torch.tensor([local_status, 1 if moved else 0], dtype=torch.int32))
    statuses = [int(q[0]) for q in seen]
    assert any(statuses) and not all(statuses), ("the scenario needs a rank that saw the timeout and one that did not", statuses)
    # the exchange is rebuilt, the next round runs and leaves bit-identical weights everywhere
    assert learner.reset_exchange()
    learner.begin_updates()
    for _ in range(3):
        learner.gradient_step_tensors(X, G, Y, sync=False)
    learner.end_updates()
    assert int(o.step_dev.item()) == start[3] + 3
    parts = gather_cpu(o.flat)
    assert all(torch.equal(parts[0], q) for q in parts[1:])
    assert not torch.equal(o.flat, start[0])
    return {"statuses_per_rank": statuses, "some_rank_stepped_before_rollback": [int(q[1]) for q in seen], "message": msg}


def scenario_vec(comm, dev, rk, world):
    """The data-parallel round of the vectorised loop: graphs of 32 updates (gather-many + 32 x two launches with the exchange
    inside) against the same updates issued one by one (gather + GraphedUpdate) -- same ids, same weights at the start."""
    import configparser
    import random
    from multiagent_gnn_policies_amd.envs import FlockParams, VecFlock
    from multiagent_gnn_policies_amd.learner.gnn_dagger import DAGGER
    from multiagent_gnn_policies_amd.learner.state_with_delay import BatchedDelayState
    from multiagent_gnn_policies_amd.learner.vec_dagger import FrameReplay, FrameUpdates, collect_round, _dp_mode
    cp = configparser.ConfigParser()
    cp['DEFAULT'] = dict(n_states='6', n_actions='2', k='3', hidden_size='32', gamma='0.99', tau='0.5', n_agents='100',
                         actor_lr='1e-3')
    cp['t'] = {}
    lanes, T, B, N, K, U = 8, 16, 20, 100, 3, 70
    torch.manual_seed(7)
    a = DAGGER(dev, cp['t'])
    torch.manual_seed(7)
    b = DAGGER(dev, cp['t'])
    assert torch.equal(a.actor_optim.flat, b.actor_optim.flat)
    assert _dp_mode(a, True) == 'p2p' and FrameUpdates.supported(a, B, N)
    p = FlockParams(n_agents=N, init_mode='grid')
    sim = VecFlock(lanes, p, dev, with_expert=True)
    st = BatchedDelayState(dev, lanes, K, 6, N)
    mem = FrameReplay(lanes, lanes * T, K, N, dev)
    np.random.seed(20 + rk)                                       # every rank collects its own episodes
    collect_round(a, sim, st, mem, torch.full((lanes,), 0.7, device=dev),
                  torch.arange(rk * lanes, (rk + 1) * lanes, dtype=torch.int32, device=dev), 3, T)
    random.seed(40 + rk)
    ids = [mem.sample_ids(B) for _ in range(U)]
    fu = FrameUpdates(a, mem, B, U, True)
    it = iter(ids)
    a.begin_updates()
    loss_a = float(fu.run_sampled(U, sampler=lambda: next(it)).item())
    a.end_updates()
    # one by one
    bufs = b.graphed_buffers(B, N)
    assert bufs is not None
    from multiagent_gnn_policies_amd import ops
    loss_b = 0.0
    b.begin_updates()
    for u in range(U):
        ops.replay_gather(mem, torch.tensor(ids[u], device=dev, dtype=torch.long), bufs[0], bufs[1], bufs[2], True)
        loss_b += b.gradient_step_tensors(*bufs)
    b.end_updates()
    assert b._graphed[B].p2p is not None
    wa, wb = a.actor_optim.flat.clone(), b.actor_optim.flat.clone()
    assert int(a.actor_optim.step_dev.item()) == U == int(b.actor_optim.step_dev.item())
    err = float((wa - wb).abs().max())
    # dense slots: the same kernels on the same operands as the one-by-one path.  Aggregated slots (the default): the K-hop
    # products are summed along the bit rows instead of over the dense slices -- fp32 re-association of the first layer's
    # input (1e-7 on a gradient entry), carried through U = 70 Adam steps of 1e-3: an entry whose gradient is itself of that
    # size moves by a fraction of a step either way (measured 6.5e-5 = 0.07 steps on the worst entry of 1,730)
    assert fu.aggregated == (os.environ.get('MGP_FRAME_AGG', '1') != '0')
    assert err <= (2e-4 if fu.aggregated else 1e-7), err
    assert abs(loss_a - loss_b) <= 1e-4 * max(1.0, abs(loss_b)), (loss_a, loss_b)
    parts = gather_cpu(wa)
    assert all(torch.equal(parts[0], q) for q in parts[1:]), "weights must be bit-identical on every rank"
    return {"updates": U, "loss_sum": loss_a, "graph_vs_single_updates_max_weight_diff": err, "bit_identical_paths": err == 0.0,
            "aggregated": fu.aggregated}


def main():
    scenario = sys.argv[1]
    rk, world, local = parallel.init_from_env()
    assert dist.is_initialized() and world > 1
    dev = torch.device('cuda', parallel.local_device_index(local))
    torch.cuda.set_device(dev)
    res = {}
    if scenario in ('train', 'vec', 'dp_timeout'):
        res = {'train': scenario_train, 'vec': scenario_vec, 'dp_timeout': scenario_dp_timeout}[scenario](None, dev, rk, world)
    else:
        comm = parallel.P2PExchange.create(1731, dev)
        assert comm is not None, "one-shot exchange did not come up"
        res = {'allreduce': scenario_allreduce, 'timeout': scenario_timeout}[scenario](comm, dev, rk, world)
        torch.cuda.synchronize()
        dist.barrier()
        comm.close()
    dist.barrier()
    if rk == 0:
        import json
        print("P2P_OK " + json.dumps(res))
    dist.destroy_process_group()


if __name__ == '__main__':
    main()
