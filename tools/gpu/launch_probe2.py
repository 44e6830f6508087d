#!/usr/bin/env python3
"""Host side of one 20-step resident launch, piece by piece: the Python above the C ABI (ctypes call stubbed out), the ctypes call
itself (argument conversion + mgp_rollout_steps_ex + the HIP launch, until it returns), and the wall time of launch + synchronize
for the bench.py path against the bare ctypes call with pre-converted arguments."""
import os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
from multiagent_gnn_policies_amd import _lib, ops

T = int(os.environ.get('PROBE_T', 20))
ro = bench.Rollout(torch.device('cuda:0'), 256, 100, 3, [32, 32], seed=1000)
ep = bench.Episodes(ro, 0, [])
ro.prepare_resident([5, T])
ro.run_resident(5)
ro.run_resident(T)
torch.cuda.synchronize()
plan = ro._plan
L = _lib.lib()
sync = torch.cuda.synchronize


def med(fn, n=60):
    v = []
    for _ in range(n):
        sync()
        v.append(fn())
    return 1e6 * float(np.median(v[5:]))


def full():
    t0 = time.perf_counter(); ep.advance(ro.run_resident, T); sync(); return time.perf_counter() - t0


def full_enqueue():
    t0 = time.perf_counter(); ep.advance(ro.run_resident, T); return time.perf_counter() - t0


class Stub(object):
    def mgp_rollout_steps_ex(self, *a):
        return 0


def py_only():
    real = plan._L
    plan._L = Stub()
    try:
        t0 = time.perf_counter(); ep.advance(ro.run_resident, T); return time.perf_counter() - t0
    finally:
        plan._L = real


state, sim = plan.state, plan.sim
rw = ro._rws[T]
fn = L.mgp_rollout_steps_ex
flags = ops.RO_ENTER_CARRY | ops.RO_EXIT_CARRY | ops.RO_SKIP_DENSE
a = (sim.x.data_ptr(), state._G[state._cur].data_ptr(), state._X[state._cur].data_ptr(), None, None, plan._cd, plan._nl, None,
     rw.data_ptr(), plan._params, sim.B, state.K, sim.N, T, plan._image_p, plan._carry_p, flags, ops._stream())


def raw():
    t0 = time.perf_counter(); fn(*a); sync(); return time.perf_counter() - t0


def raw_enqueue():
    t0 = time.perf_counter(); fn(*a); return time.perf_counter() - t0


def sync_only():
    t0 = time.perf_counter(); sync(); return time.perf_counter() - t0


e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); e1.record(); sync()


def overhead(call, n=60):
    """median over n launches of (wall time of launch + synchronize) - (the SAME launch's own begin -> end, stamped by the launch:
    mgp_set_launch_events): the flock keeps evolving from launch to launch, so only differences within one launch compare"""
    ov, kn, en = [], [], []
    for _ in range(n):
        sync()
        L.mgp_set_launch_events(e0.cuda_event, e1.cuda_event)
        t0 = time.perf_counter(); call(); t1 = time.perf_counter(); sync(); t2 = time.perf_counter()
        k = e0.elapsed_time(e1) * 1e3
        ov.append(1e6 * (t2 - t0) - k); kn.append(k); en.append(1e6 * (t1 - t0))
    return float(np.median(ov[5:])), float(np.median(kn[5:])), float(np.median(en[5:]))


ob, kb, eb = overhead(lambda: ep.advance(ro.run_resident, T))
orw, kr, er = overhead(lambda: fn(*a))
print('T=%d  kernel (launch-stamped events) %.1f us' % (T, kb))
print('bench path (Episodes.advance -> Rollout.run_resident -> ResidentPlan.run): wall - kernel = %.1f us per launch | the call returns after %.1f us | Python above the C ABI %.1f us' % (ob, eb, med(py_only)))
print('bare ctypes call, arguments pre-converted: wall - kernel = %.1f us per launch | the call returns after %.1f us' % (orw, er))
print('synchronize on an idle device %.1f us' % med(sync_only))
