"""GPU: the one-shot gradient exchange (mgp_p2p_*, csrc/p2p_device.h) between TWO PROCESSES.  hipIpc maps memory between
processes on one device, so the single MI355X of the test box runs the real thing: both ranks' kernels run concurrently on
the GPU and exchange 64-bit {sequence | fp32} packets through each other's mailboxes.  The process group (gloo here) only
carries the 64-byte IPC handles and the reference values; on a multi-GPU node the same code runs one rank per device.
Scenarios live in tests/p2p_worker.py."""
import json
import os
import socket
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _port():
    s = socket.socket(); s.bind(('127.0.0.1', 0)); p = s.getsockname()[1]; s.close()
    return p


def run_ranks(scenario, world=2, timeout=600, **extra):
    port = str(_port())
    procs = []
    for rk in range(world):
        env = dict(os.environ, PYTHONPATH=ROOT, RANK=str(rk), LOCAL_RANK=str(rk), WORLD_SIZE=str(world),
                   MASTER_ADDR='127.0.0.1', MASTER_PORT=port, MGP_DIST_BACKEND='gloo', HSA_ENABLE_IPC_MODE_LEGACY='0',
                   MGP_P2P_TIMEOUT_MS='60000')     # the ranks share ONE device here and take turns on it: on a loaded box an exchange
                                                   # of eight ranks has waited out the 5 s default (the 'timeout' scenario sets its own)
        env.update(extra)
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, 'tests', 'p2p_worker.py'), scenario], cwd=ROOT,
                                      env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = []
    try:
        for p in procs:
            outs.append(p.communicate(timeout=timeout))
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    for p, (o, e) in zip(procs, outs):
        assert p.returncode == 0, (o[-2000:], e[-4000:])
    line = [l for l in outs[0][0].splitlines() if l.startswith('P2P_OK ')]
    assert line, outs[0]
    return json.loads(line[0][len('P2P_OK '):])


def test_two_processes_exchange_exactly_in_rank_order_and_inside_a_graph():
    r = run_ranks('allreduce')
    assert r['exchanges'] >= 40 + 6 + 32 * 13 and r['exchange_us_in_graph'] < 200.0, r


def test_three_ranks_on_one_device():
    r = run_ranks('allreduce', world=3)
    assert r['exchanges'] > 0


def test_eight_ranks_on_one_device():
    """MGP_P2P_MAX_WORLD = 8 = the node north_star names: eight processes, each with its own mailbox of 8 x 2 x 1,731 packets
    (216 KB), every rank pushing into seven peers and polling seven sources per entry.  On the one-GPU box the eight ranks'
    kernels share the device (an upper bound on the exchange's latency, a real check of its protocol at full width)."""
    r = run_ranks('allreduce', world=8, timeout=900)
    assert r['exchanges'] >= 40 + 6 + 32 * 13 and r['exchange_us_in_graph'] < 400.0, r


def test_data_parallel_update_at_eight_ranks_is_bit_identical_on_every_rank():
    r = run_ranks('train', world=8, timeout=900)
    assert r['max_weight_diff_vs_single_process'] <= 1e-7


def test_missing_peer_is_a_status_not_a_hang():
    r = run_ranks('timeout')
    assert r['status'] == 1


@pytest.mark.parametrize('world', [2, 3])
def test_timeout_inside_a_data_parallel_round_rolls_every_rank_back(world):
    """DAGGER.begin_updates() .. end_updates() with one rank two seconds late in the middle of the round (exchange timeout 400 ms):
    the early ranks see the timeout, the late one does not -- end_updates() raises on ALL of them with weights, Adam moments
    and step counters restored to the round's start (identical across ranks), and after reset_exchange() the next round runs."""
    r = run_ranks('dp_timeout', world=world)
    assert any(r['statuses_per_rank']) and not all(r['statuses_per_rank']), r
    assert any(r['some_rank_stepped_before_rollback']), r
    assert 'restored to the start of the round' in r['message']


def test_data_parallel_update_with_the_exchange_inside_equals_the_averaged_single_process_update():
    r = run_ranks('train')
    assert r['max_weight_diff_vs_single_process'] <= 1e-7


@pytest.mark.parametrize('agg', ['1', '0'], ids=['aggregated', 'dense'])
def test_data_parallel_round_of_32_update_graphs_equals_single_updates(agg):
    """70 data-parallel updates as graphs of 32 (the exchange inside every update's second launch) against the same updates
    issued one by one on dense gathered minibatches: the dense slots reproduce them to the bit (<= 1e-7), the aggregated slots
    (mgp_replay_aggregate + mgp_train_step_agg with the exchange) to fp32 re-association of the K-hop sums carried through 70
    Adam steps (a fifth of one step on the worst entry); the loss sums agree to 1e-4 either way, every rank ends bit-identical."""
    r = run_ranks('vec', MGP_FRAME_AGG=agg)
    assert r['aggregated'] == (agg == '1')
    assert r['graph_vs_single_updates_max_weight_diff'] <= (2e-4 if agg == '1' else 1e-7)
