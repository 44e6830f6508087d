"""GPU: the drop-in loops end to end -- gym-style env facade against the oracle spec, and train.py on a tiny cfg."""
import configparser
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from conftest import ROOT
from oracle import flock as ofl

pytestmark = pytest.mark.gpu


def _args(**kw):
    cp = configparser.ConfigParser()
    base = dict(alg='dagger', k='3', n_agents='30', comm_radius='1.0', v_max='3.0', dt='0.01', n_states='6',
                n_actions='2')
    base.update({k: str(v) for k, v in kw.items()})
    cp['DEFAULT'] = base
    cp['t'] = {}
    return cp['t']


@pytest.mark.parametrize('env_id,variant', [('FlockingRelative-v0', {}), ('FlockingLeader-v0', {'n_leaders': 2}),
                                            ('FlockingTwoFlocks-v0', {'two_flocks': True}),
                                            ('FlockingStochastic-v0', {'link_drop': 0.1, 'link_seed': 7})])
def test_env_facade_matches_oracle_spec(env_id, variant):
    from multiagent_gnn_policies_amd import envs
    n = 30
    env = envs.make(env_id, max_episode_steps=6)
    assert isinstance(env.env, envs.FlockingRelativeEnv)
    env.env.params_from_cfg(_args(n_agents=n))
    env.seed(7)
    vals, net = env.reset()
    p = ofl.FlockParams(n_agents=n, **variant)
    x = ofl.reset(np.random.RandomState(7), p)                 # same RNG stream, same spec
    h = ofl.helpers(x, p)
    assert vals.shape == (n, 6) and net.shape == (n, n) and vals.dtype == np.float64
    assert np.sum(np.diag(net)) == 0                              # state_with_delay.py:26
    assert np.array_equal(net, h['network'])
    assert np.max(np.abs(vals - h['values']) / np.maximum(1, np.abs(h['values']))) <= 1e-11
    done, steps = False, 0
    while not done:
        ud = env.env.controller(False)
        assert np.max(np.abs(ud - ofl.controller(x, p, centralized=False))) <= 1e-11
        u = env.env.controller()                                      # default: the global teacher (p.centralized)
        uo = ofl.controller(x, p)
        assert u.shape == (n, 2) and p.centralized
        assert np.max(np.abs(u - uo)) <= 1e-10
        uc = env.env.controller(True)
        assert np.max(np.abs(uc - ofl.controller(x, p, centralized=True))) <= 1e-10
        (vals, net), r, done, info = env.step(u)
        x, v2, n2, r2 = ofl.step(x, u.astype(np.float32), p)        # the device simulator consumes actions as fp32
        assert np.array_equal(net, n2)
        assert np.max(np.abs(vals - v2) / np.maximum(1, np.abs(v2))) <= 1e-11
        assert abs(r - r2) <= 1e-12 * max(1.0, abs(r2))
        steps += 1
    assert steps == 6 and info.get('TimeLimit.truncated')
    env.close()


def test_train_py_smoke_cfg():
    """python train.py cfg/smoke.cfg: header once, then `section, mean, std` per section (reference train.py:51-60)."""
    env = dict(os.environ, PYTHONPATH=ROOT)
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'train.py'), 'cfg/smoke.cfg'], cwd=ROOT, env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.strip().splitlines() if l.strip()]
    assert lines[0] == 'alg, reward'
    rows = {l.split(',')[0].strip(): l for l in lines[1:]}
    assert set(rows) == {'dagger', 'cloning', 'baseline'}
    for name, l in rows.items():
        parts = [s.strip() for s in l.split(',')]
        mean, std = float(parts[1]), float(parts[2])
        assert np.isfinite(mean) and mean < 0 and std >= 0           # reward = -velocity variance
    # the expert baseline must not be worse than an untrained imitation policy by construction of the task
    assert float(rows['baseline'].split(',')[1]) <= 0


def test_dagger_learning_reduces_loss():
    """A few hundred DAGGER updates on expert-labelled states of the device simulator drive the imitation loss down."""
    from multiagent_gnn_policies_amd.envs import FlockParams, VecFlock
    from multiagent_gnn_policies_amd.learner.gnn_dagger import DAGGER
    from multiagent_gnn_policies_amd.learner.state_with_delay import BatchedDelayState
    cp = configparser.ConfigParser()
    cp['DEFAULT'] = dict(n_states='6', n_actions='2', k='3', hidden_size='32', gamma='0.99', tau='0.5', n_agents='40',
                         actor_lr='1e-3')
    cp['t'] = {}
    torch.manual_seed(0)
    dev = torch.device('cuda:0')
    learner = DAGGER(dev, cp['t'])
    B, N = 16, 40
    sim = VecFlock(B, FlockParams(n_agents=N, init_mode='grid'), dev, with_expert=True)
    sim.reset(np.random.RandomState(0))
    st = BatchedDelayState(dev, B, 3, 6, N)
    Xs, Gs, Ys = [], [], []
    for t in range(12):
        st.push(sim.network, sim.features)
        Xs.append(st.delay_state.clone()); Gs.append(st.delay_gso.clone())
        Ys.append(sim.controller().permute(0, 2, 1).reshape(B, 1, 2, N).contiguous().clone())
        sim.step(sim.controller())
    X, G, Y = torch.cat(Xs), torch.cat(Gs), torch.cat(Ys)
    losses = []
    g = torch.Generator().manual_seed(0)
    for it in range(300):
        idx = torch.randint(0, X.shape[0], (32,), generator=g).to(dev)
        losses.append(learner.gradient_step_tensors(X[idx], G[idx], Y[idx]))
    assert np.isfinite(losses).all()
    assert np.mean(losses[-30:]) < 0.7 * np.mean(losses[:30])


def _torchrun(script_args, timeout=900):
    import socket
    sock = socket.socket(); sock.bind(('127.0.0.1', 0)); port = sock.getsockname()[1]; sock.close()
    env = dict(os.environ, PYTHONPATH=ROOT, MGP_DIST_BACKEND='gloo')        # two ranks share the one GPU of the box
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2',
           '--master-addr', '127.0.0.1', '--master-port', str(port)] + script_args
    return subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                          timeout=timeout)


def test_train_py_two_ranks_data_parallel():
    """torchrun x2: episodes dealt to ranks, flat-gradient all-reduce per update, rank 0 prints the sweep table."""
    r = _torchrun([os.path.join(ROOT, 'train.py'), 'cfg/smoke.cfg'])
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.strip().splitlines() if l.strip() and ',' in l]
    assert lines[0] == 'alg, reward'
    assert [l.split(',')[0] for l in lines[1:]] == ['dagger', 'cloning', 'baseline']      # printed once (rank 0)


def test_fast_loop_mode_is_bit_identical_to_the_numpy_loop():
    """The one-environment loops of this package run the gym-style env in its fast loop mode (device actions, device rewards,
    fp32-only observations): same trajectory, same rewards, same replay labels as the gym_flock-compatible numpy mode."""
    import random
    from multiagent_gnn_policies_amd import envs
    from multiagent_gnn_policies_amd.learner.gnn_dagger import DAGGER
    from multiagent_gnn_policies_amd.learner.imitation import ImitationRun
    from multiagent_gnn_policies_amd.learner.rollouts import PolicyRunner, run_episode
    args = _args(n_agents=30, hidden_size=16, gamma=0.99, tau=0.5, actor_lr=1e-3, batch_size=4, buffer_size=100,
                 updates_per_step=2, n_train_episodes=1, test_interval=1, n_test_episodes=1, debug=False, env='FlockingRelative-v0')
    res = []
    for fast in (False, True):
        random.seed(2); np.random.seed(2); torch.manual_seed(2)
        env = envs.make('FlockingRelative-v0', max_episode_steps=12)
        env.env.params_from_cfg(args)
        env.seed(2)
        learner = DAGGER('cuda:0', args)
        run = ImitationRun(env, learner, args, torch.device('cuda:0'))
        assert run.fast                                          # the loop switches the env to the fast mode itself ...
        if not fast:
            env.env.fast_loop = False; run.fast = False          # ... the numpy mode is what a gym_flock user gets
        run.collect(0.5)
        labels = torch.cat([t.action for t in run.memory.buffer]).cpu().numpy()
        rewards = torch.cat([t.reward for t in run.memory.buffer]).cpu().numpy()
        last = run.memory.buffer[-1].next_state
        ep = run_episode(env, PolicyRunner(learner, 'cuda:0', args).act)
        res.append((labels, rewards, last.delay_gso.cpu().numpy(), last.delay_state.cpu().numpy(), ep))
        assert type(ep) is float
    for a, b in zip(res[0][:4], res[1][:4]):
        assert np.array_equal(a, b)
    assert res[0][4] == res[1][4]
    env = envs.make('FlockingRelative-v0', max_episode_steps=3)
    env.env.params_from_cfg(args); env.env.fast_loop = True
    obs = env.reset()
    with pytest.raises(Exception):
        np.asarray(obs[0])                                       # fp32 device side only: asking for numpy fails loudly


def test_train_py_dagger_vec_single_and_two_ranks():
    """`alg = dagger_vec` through train.py: the device-collecting vectorised loop, single process and torchrun x2 (episodes
    and coin streams are functions of the GLOBAL episode index; the data-parallel update goes through GraphedUpdate)."""
    env = dict(os.environ, PYTHONPATH=ROOT)
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'train.py'), 'cfg/smoke_vec.cfg'], cwd=ROOT, env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    rows = [l for l in r.stdout.strip().splitlines() if l.startswith('dagger_vec')]
    assert len(rows) == 1 and np.isfinite(float(rows[0].split(',')[1])) and float(rows[0].split(',')[1]) < 0
    r2 = _torchrun([os.path.join(ROOT, 'train.py'), 'cfg/smoke_vec.cfg'])
    assert r2.returncode == 0, r2.stderr[-3000:]
    rows2 = [l for l in r2.stdout.strip().splitlines() if l.startswith('dagger_vec')]
    assert len(rows2) == 1 and np.isfinite(float(rows2[0].split(',')[1]))


def test_bench_two_ranks_contract():
    """bench.py under the driver's multi-GPU launch line (2 ranks): one JSON line, weak scaling, aggregate value."""
    import json
    r = _torchrun([os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '20', '--warmup', '4', '--episodes', '32'])
    assert r.returncode == 0, r.stderr[-3000:]
    js = [l for l in r.stdout.splitlines() if l.startswith('{')]
    assert len(js) == 1
    d = json.loads(js[0])
    assert d['n_gpus'] == 2 and d['scaling'] == 'weak' and d['steps'] == 20
    assert d['config']['episodes_total'] == 64
    assert abs(d['value'] - 64 * 100 * 20 / (d['ms_per_step'] * 20 / 1e3)) / d['value'] < 1e-6
    assert 'cpu_baseline' not in d and 'roofline' in d


def _bench_no_launcher(extra, backend='gloo', timeout=900):
    """`python bench.py --gpus 2 ...` exactly as typed -- NO torchrun, WORLD_SIZE unset: bench.py starts its own ranks."""
    env = dict(os.environ, PYTHONPATH=ROOT)
    for k_ in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT', 'MGP_DIST_BACKEND'):
        env.pop(k_, None)
    if backend is not None:
        env['MGP_DIST_BACKEND'] = backend
        env['MGP_P2P_TIMEOUT_MS'] = '60000'                  # gloo here = two ranks taking turns on ONE device
    return subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2'] + extra, cwd=ROOT, env=env,
                          stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=timeout)


def test_bench_gpus_2_without_a_launcher_starts_two_ranks():
    import json
    r = _bench_no_launcher(['--steps', '20', '--warmup', '4', '--episodes', '32', '--no-roofline'])
    assert r.returncode == 0, r.stderr[-3000:]
    js = [l for l in r.stdout.splitlines() if l.startswith('{')]
    assert len(js) == 1
    d = json.loads(js[0])
    assert d['n_gpus'] == 2 and d['dist'] == {'backend': 'gloo', 'world_size': 2, 'devices_visible': torch.cuda.device_count()}
    assert d['config']['episodes_total'] == 64 and d['parity']['ok']


def test_bench_refuses_more_rccl_ranks_than_devices():
    """The driver's `python bench.py --gpus 8` on a box with fewer devices must fail loudly, not print n_gpus: 1."""
    if torch.cuda.device_count() >= 2:
        pytest.skip("needs a box with fewer devices than ranks")
    r = _bench_no_launcher(['--steps', '20', '--warmup', '4'], backend=None)
    assert r.returncode != 0 and 'one GPU per rank' in r.stderr
    assert not [l for l in r.stdout.splitlines() if l.startswith('{')]


@pytest.mark.parametrize('gpus', [1, 2])
def test_bench_dagger_round(gpus):
    """bench.py --dagger (BASELINE configs[3]): collection + graph-captured updates; with 2 ranks (no launcher, sharing the GPU)
    the gradient goes through the one-shot exchange and the ranks' weights end bit-identical."""
    import json
    extra = ['--dagger', '--steps', '40', '--warmup', '8', '--episodes', '16', '--updates', '100']
    if gpus == 2:
        r = _bench_no_launcher(extra)
    else:
        r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py')] + extra, cwd=ROOT, env=dict(os.environ, PYTHONPATH=ROOT),
                           stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith('{')][0])
    assert d['n_gpus'] == gpus and d['weights_bit_identical_across_ranks'] is True
    assert d['updates']['count'] == 100 and np.isfinite(d['updates']['mean_loss']) and d['value'] > 0
    assert d['updates']['exchange'] == ('p2p' if gpus == 2 else 'none (single process)')
    assert d['dist']['world_size'] == gpus


@pytest.mark.parametrize('n,k,paths', [(100, 3, {'two_launch', 'resident', 'resident_grid'}), (200, 4, {'two_launch', 'resident'}),
                                       (300, 3, {'two_launch', 'factored'})])
def test_bench_single_gpu_contract_and_paths(n, k, paths):
    """python bench.py: one JSON line with the contract's keys; the step implementations timed for the shape."""
    import json
    env = dict(os.environ, PYTHONPATH=ROOT)
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--steps', '30', '--warmup', '5', '--episodes', '16',
                        '--agents', str(n), '--taps', str(k), '--no-cpu-baseline'], cwd=ROOT, env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    js = [l for l in r.stdout.splitlines() if l.startswith('{')]
    assert len(js) == 1
    d = json.loads(js[0])
    for key in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling',
                'vs_baseline', 'dtype', 'data', 'config', 'roofline'):
        assert key in d, key
    if 'resident' in paths:
        paths = paths | {'resident_dense_exit'}
    assert set(d['paths']) == paths and d['config']['state_finite']
    headline = 'factored' if 'factored' in paths else ('resident' if 'resident' in paths else 'two_launch')   # by shape
    assert d['value'] == d['paths'][headline]['value']
    assert d['parity']['ok'] and d['parity']['tol'] == 1e-5 and {'max_abs', 'max_rel'} <= set(d['parity'])
    assert {'bound', 'achieved', 'peak', 'unit', 'frac', 'traffic'} <= set(d['roofline'])
    if 'resident' in paths:
        # the resident region is timed twice over the same steps: plain launch = value, event-stamped pass = the roofline's duration
        r_ = d['paths']['resident']
        assert r_['ms_per_step'] == d['ms_per_step'] and r_['ms_per_step_event_pass'] > 0
        assert 0 < r_['launch_ms_hip_events'] <= 1.05 * r_['ms_per_step_event_pass'] * d['steps']
        assert abs(d['roofline']['avg_launch_ms'] - r_['launch_ms_hip_events']) <= 1e-9 + 1e-6 * r_['launch_ms_hip_events']
        # [r5] the line says what the timed kernel defers and what bounds it: the dense slices (RO_SKIP_DENSE) with the same region
        # timed with them rebuilt inside the launch, vector-issue as the bound with the matrix figure beside it, a median of repeats
        assert 'MGP_RO_SKIP_DENSE' in d['config']['step_path'] and 'resident_dense_exit' in d['config']['step_path']
        assert d['paths']['resident_dense_exit']['ms_per_step'] > 0.9 * r_['ms_per_step']
        # [r6] the vector-issue figure rests on a counter pass + ISA mix of the headline shape: other shapes carry the algorithmic matrix-pipe
        # figure as `frac` and say that their limiter was not measured
        assert d['roofline']['bound'] == ('valu' if (n, k) == (100, 3) else 'mfma') and {'achieved', 'peak', 'frac'} <= set(d['roofline']['mfma'])
        assert d['roofline']['frac'] is not None and d['roofline']['frac'] > 0
        if (n, k) == (100, 3):
            assert 2.0 < d['roofline']['mean_cycles_per_valu_instruction'] < 5.0
        vm = d['value_median_of']
        assert vm['n'] >= 3 and len(vm['samples_ms_per_step']) == vm['n'] and vm['samples_ms_per_step'][0] == d['ms_per_step']
        assert min(vm['samples_ms_per_step']) <= vm['ms_per_step'] <= max(vm['samples_ms_per_step'])


def test_device_replay_ring_and_sampling():
    from multiagent_gnn_policies_amd.learner.vec_dagger import DeviceReplay
    import random
    rb = DeviceReplay(10, 2, 3, 4, 2, torch.device('cuda:0'))
    for start in (0, 4, 8):                                    # 12 inserts into a ring of 10
        ids = torch.arange(start, start + 4, device='cuda').float()
        rb.insert_batch(ids.view(4, 1, 1, 1).expand(4, 2, 3, 4).contiguous(),
                        ids.view(4, 1, 1, 1).expand(4, 2, 4, 4).contiguous(),
                        ids.view(4, 1, 1, 1).expand(4, 1, 2, 4).contiguous())
    assert rb.curr_size == 10 and rb.position == 2
    stored = sorted(rb.delay_state[:, 0, 0, 0].cpu().tolist())
    assert stored == [2, 3, 4, 5, 6, 7, 8, 9, 10, 11]           # the two oldest were overwritten
    random.seed(0)
    xs, gs, ys = rb.sample(10)
    assert sorted(xs[:, 0, 0, 0].cpu().tolist()) == stored       # without replacement
    assert torch.equal(xs[:, 0, 0, 0], gs[:, 0, 0, 0]) and torch.equal(xs[:, 0, 0, 0], ys[:, 0, 0, 0])
    with pytest.raises(ValueError):
        rb.sample(11)


def test_indexed_updates_equal_the_sampled_path():
    """mgp_train_step_indexed (gather inside the kernel, device-side cursor, one graph replay per update) against the
    same updates through DeviceReplay.sample + GraphedUpdate: identical weights and losses, update by update."""
    import random
    from multiagent_gnn_policies_amd.learner.gnn_dagger import DAGGER
    from multiagent_gnn_policies_amd.learner.vec_dagger import DeviceReplay, IndexedUpdates
    dev = torch.device('cuda:0')
    N, K, B, U, cap = 100, 3, 20, 12, 64
    args = _args(n_agents=N, k=K, hidden_size=32, gamma=0.99, tau=0.5, actor_lr=1e-3)
    rb = DeviceReplay(cap, K, 6, N, 2, dev)
    g = torch.Generator(device='cuda').manual_seed(3)
    for _ in range(cap // 16):
        rb.insert_batch(torch.randn((16, K, 6, N), device=dev, generator=g),
                        0.05 * torch.rand((16, K, N, N), device=dev, generator=g),
                        torch.randn((16, 1, 2, N), device=dev, generator=g))
    random.seed(5)
    ids = [random.sample(range(rb.curr_size), B) for _ in range(U)]
    torch.manual_seed(11); a = DAGGER(dev, args)
    torch.manual_seed(11); b = DAGGER(dev, args)
    assert IndexedUpdates.supported(a, B, N)
    iu = IndexedUpdates(a, rb, B, 16)
    total = iu.run(ids[:5]).item() + iu.run(ids[5:]).item()        # two rounds: the cursor restarts, the step counter goes on
    losses = []
    for row in ids:
        idx = torch.tensor(row, device=dev)
        losses.append(b.gradient_step_tensors(rb.delay_state[idx], rb.delay_gso[idx], rb.action[idx]))
    assert a.actor_optim.step_count == b.actor_optim.step_count == U
    assert torch.equal(a.actor_optim.flat, b.actor_optim.flat)
    assert abs(total - sum(losses)) <= 1e-5 * max(1.0, abs(sum(losses)))
    assert torch.allclose(iu.loss_hist[:7].cpu(), torch.tensor(losses[5:]), rtol=0, atol=1e-6)


@pytest.mark.parametrize('kind', ['dense', 'frames'])
def test_pipelined_sampling_equals_the_uploaded_round(kind):
    """`run_sampled` (the update path of train_dagger_vec: pinned double-buffered index staging, graphs of 32 updates + a
    remainder on the one-update graph) against `run(ids)` on the same ids, U not a multiple of 32: identical weights and
    per-update losses -- for the dense replay (IndexedUpdates) and the frame replay (FrameUpdates: gather-many slots)."""
    import random
    from multiagent_gnn_policies_amd.envs import FlockParams, VecFlock
    from multiagent_gnn_policies_amd.learner.gnn_dagger import DAGGER
    from multiagent_gnn_policies_amd.learner.state_with_delay import BatchedDelayState
    from multiagent_gnn_policies_amd.learner.vec_dagger import (DeviceReplay, IndexedUpdates, FrameReplay, FrameUpdates,
                                                                collect_round, UPDATES_PER_GRAPH)
    dev = torch.device('cuda:0')
    N, K, B, U = 100, 3, 20, 2 * UPDATES_PER_GRAPH + 7
    args = _args(n_agents=N, k=K, hidden_size=32, gamma=0.99, tau=0.5, actor_lr=1e-3)
    torch.manual_seed(11); a = DAGGER(dev, args)
    torch.manual_seed(11); b = DAGGER(dev, args)
    if kind == 'dense':
        mem = DeviceReplay(64, K, 6, N, 2, dev)
        g = torch.Generator(device='cuda').manual_seed(3)
        for _ in range(4):
            mem.insert_batch(torch.randn((16, K, 6, N), device=dev, generator=g),
                             0.05 * torch.rand((16, K, N, N), device=dev, generator=g),
                             torch.randn((16, 1, 2, N), device=dev, generator=g))
        random.seed(5)
        ids = [random.sample(range(mem.curr_size), B) for _ in range(U)]
        ua, ub = IndexedUpdates(a, mem, B, U), IndexedUpdates(b, mem, B, U)
    else:
        lanes, T = 8, 12
        sim = VecFlock(lanes, FlockParams(n_agents=N, init_mode='grid'), dev, with_expert=True)
        st = BatchedDelayState(dev, lanes, K, 6, N)
        mem = FrameReplay(lanes, lanes * T, K, N, dev)
        np.random.seed(2)
        collect_round(a, sim, st, mem, torch.full((lanes,), 0.6, device=dev), torch.arange(lanes, dtype=torch.int32, device=dev), 1, T)
        random.seed(5)
        ids = [mem.sample_ids(B) for _ in range(U)]
        ua, ub = FrameUpdates(a, mem, B, U, True), FrameUpdates(b, mem, B, U, True)
    it = iter(ids)
    la = float(ua.run_sampled(U, sampler=lambda: next(it)).item())
    lb = float(ub.run(ids).item())
    assert a.actor_optim.step_count == b.actor_optim.step_count == U
    assert int(a.actor_optim.step_dev.item()) == U
    assert torch.equal(a.actor_optim.flat, b.actor_optim.flat)
    assert torch.equal(ua.loss_hist[:U], ub.loss_hist[:U]) and la == lb


def test_borrowed_env_is_handed_back_in_numpy_mode():
    """This package's loops run the user's gym-style env in its fast loop mode and must hand it back as they found it: a
    gym_flock-style caller keeps getting numpy-convertible observations and float rewards afterwards."""
    from multiagent_gnn_policies_amd import envs
    from multiagent_gnn_policies_amd.learner.gnn_dagger import DAGGER
    from multiagent_gnn_policies_amd.learner.rollouts import policy_episode_reward, fast_loop_mode
    dev = torch.device('cuda:0')
    args = _args(n_agents=20, k=2, hidden_size=16, gamma=0.99, tau=0.5, actor_lr=1e-3, comm_radius=1.0, v_max=3.0)
    env = envs.make('FlockingRelative-v0', device='cuda:0', max_episode_steps=6)
    env.env.params_from_cfg(args)
    env.seed(3)
    assert env.env.fast_loop is False
    r = policy_episode_reward(env, DAGGER(dev, args), dev, args)
    assert np.isfinite(r) and env.env.fast_loop is False
    obs = env.reset()
    assert np.asarray(obs[0]).shape == (20, 6) and np.asarray(obs[1]).shape == (20, 20)
    obs, rew, done, _ = env.step(np.zeros((20, 2)))
    assert isinstance(rew, float) and np.asarray(obs[0]).dtype == np.float64
    # a mid-episode toggle continues the episode, and fast-mode observations answer hasattr() instead of raising
    x_before = env.env._sim.x.clone()
    with fast_loop_mode(env) as fast:
        assert fast and torch.equal(env.env._sim.x, x_before)
        o2, r2, _, _ = env.step(torch.zeros((20, 2), device=dev))
        assert torch.is_tensor(r2) and not hasattr(o2[0], 'transpose')
        with pytest.raises(Exception):
            np.asarray(o2[0])
    assert env.env.fast_loop is False


def test_vectorised_dagger_trains():
    """Device-resident DAGGER on 16 parallel episodes: finite statistics, updates happened, policy improves over
    the untrained network on the imitation loss of fresh expert-labelled states."""
    from multiagent_gnn_policies_amd.learner.vec_dagger import train_dagger_vec
    cp = configparser.ConfigParser()
    cp['DEFAULT'] = dict(alg='dagger', batch_size='32', buffer_size='2000', updates_per_step='4', seed='1',
                         actor_lr='1e-3', n_train_episodes='32', beta_coeff='0.993', test_interval='40',
                         n_test_episodes='4', k='3', hidden_size='32', gamma='0.99', tau='0.5',
                         env='FlockingRelative-v0', v_max='3.0', comm_radius='1.0', n_agents='40', n_actions='2',
                         n_states='6', debug='False', dt='0.01', init_mode='grid')
    cp['t'] = {}
    import random
    random.seed(1); np.random.seed(1); torch.manual_seed(1)
    stats = train_dagger_vec(cp['t'], 'cuda:0', n_envs=16, episode_steps=40)
    assert np.isfinite(stats['mean']) and stats['mean'] < 0 and stats['std'] >= 0
    assert stats['updates'] == 2 * 4 * 16
    # rollouts on the collecting resident kernel, compact frames in the replay
    assert stats['collect'] == 'device' and stats['replay_bytes_per_transition'] == 48 * 40 + 4


def test_eval_model_shipped_checkpoint_flocks():
    """§8(f)-4: the checkpoint evaluation harness (reference test_model.py:14-47).  The reference's shipped K=3 policy,
    trained against gym-flock, must flock in this simulator: far better than doing nothing, on both evaluation paths."""
    import eval_model
    from multiagent_gnn_policies_amd import envs
    from multiagent_gnn_policies_amd.learner.rollouts import run_episode
    cp = configparser.ConfigParser()
    cp.read(os.path.join(ROOT, 'cfg', 'flocking_dagger_n100_k3.cfg'))
    cp['test']['n_test_episodes'] = '2'
    args = cp['test']
    ckpt = os.path.join(ROOT, eval_model.DEFAULT_ACTOR)
    one_env = eval_model.evaluate_section(args, ckpt, verbose=False)
    again = eval_model.evaluate_section(args, ckpt, verbose=False)
    assert one_env == again                                            # seeded: bit-reproducible
    cp['test']['n_test_episodes'] = '64'
    lanes = eval_model.evaluate_section(args, ckpt, lanes=32)
    assert len(one_env) == 2 and len(lanes) == 64
    env = envs.make(args.get('env'))
    env.env.params_from_cfg(args)
    env.seed(args.getint('seed'))
    idle = run_episode(env, lambda _o: np.zeros((100, 2)))
    teacher = run_episode(env, lambda _o: env.env.controller())
    assert idle < -1500
    assert max(one_env) < 0 and min(one_env) > 0.1 * idle             # rewards are negative costs
    assert np.mean(lanes) > 0.1 * idle
    assert teacher > 0.1 * idle                                       # the default (global) teacher flocks too


def test_vectorised_test_episodes_match_the_sequential_loop():
    """train_dagger's test phase runs its episodes side by side on the episode-resident kernel; same reset draws as
    the one-at-a-time loop (reference gnn_dagger.py:194-203), rewards equal up to closed-loop fp32 rounding."""
    import eval_model
    from multiagent_gnn_policies_amd import envs
    from multiagent_gnn_policies_amd.learner.gnn_dagger import DAGGER
    from multiagent_gnn_policies_amd.learner.rollouts import policy_episode_reward, policy_episode_rewards
    cp = configparser.ConfigParser()
    cp.read(os.path.join(ROOT, 'cfg', 'flocking_dagger_n100_k3.cfg'))
    args = cp['test']
    dev = torch.device('cuda:0')
    learner = DAGGER(dev, args)
    learner.load_model(os.path.join(ROOT, eval_model.DEFAULT_ACTOR), dev)
    out = []
    for vectorised in (True, False):
        env = envs.make(args.get('env'), max_episode_steps=60)
        env.env.params_from_cfg(args)
        env.seed(5)
        if vectorised:
            out.append(policy_episode_rewards(env, learner, dev, args, 3))
        else:
            out.append([policy_episode_reward(env, learner, dev, args) for _ in range(3)])
        env.close()
    a, b = np.asarray(out[0]), np.asarray(out[1])
    assert a.shape == b.shape == (3,)
    # same reset draws, same policy; the two aggregation orders differ in fp32 rounding, which 60 closed-loop steps
    # amplify to ~5e-4 of the episode reward (measured; test_gpu_rollout.py holds the per-step exactness checks)
    assert np.all(np.abs(a - b) <= 3e-3 * np.abs(b)), (a, b)


def test_vectorised_dagger_learns_to_flock():
    """End to end: 32 episodes of vectorised DAGGER (16 lanes x 150 steps, 1,280 updates) must yield a policy that
    flocks -- regression test for the replay ring keeping whole episodes (a ring shorter than one lock-step round holds
    only already-flocked states and the policy never sees a start-up state: reward -4,300 instead of -67 at N = 100)."""
    import random
    from multiagent_gnn_policies_amd import envs
    from multiagent_gnn_policies_amd.learner.rollouts import run_episode
    from multiagent_gnn_policies_amd.learner.vec_dagger import train_dagger_vec
    cp = configparser.ConfigParser()
    cp['DEFAULT'] = dict(alg='dagger', batch_size='20', buffer_size='1000', updates_per_step='40', seed='1',
                         actor_lr='5e-4', n_train_episodes='32', beta_coeff='0.993', test_interval='40',
                         n_test_episodes='8', k='3', hidden_size='32', gamma='0.99', tau='0.5',
                         env='FlockingRelative-v0', v_max='3.0', comm_radius='1.0', n_agents='40', n_actions='2',
                         n_states='6', debug='False', dt='0.01', init_mode='grid')
    cp['t'] = {}
    random.seed(1); np.random.seed(1); torch.manual_seed(1)
    stats = train_dagger_vec(cp['t'], 'cuda:0', n_envs=16, episode_steps=150)
    assert stats['updates'] == 2 * 40 * 16
    env = envs.make('FlockingRelative-v0', max_episode_steps=150)
    env.env.params_from_cfg(cp['t'])
    env.seed(5)
    idle = np.mean([run_episode(env, lambda _o: np.zeros((40, 2))) for _ in range(3)])
    assert idle < -500
    assert stats['mean'] > 0.25 * idle, (stats['mean'], idle)          # measured: -66 vs -860


@pytest.mark.parametrize('kw,B', [(dict(n_agents=100), 5), (dict(n_agents=100), 1), (dict(n_agents=100, two_flocks=True), 3),
                                  (dict(n_agents=50, n_leaders=2, min_degree=3), 4), (dict(n_agents=150, init_mode='disc'), 2),
                                  (dict(n_agents=200), 3)])
def test_batched_reset_sampler_is_the_sequential_one(kw, B):
    """envs/flocking.py::sample_initial_states (candidates built from one block of the generator's uniforms, acceptance
    statistics from mgp_flock_reset_check) against B sequential sample_initial_state calls and against the oracle's sampler
    (oracle/flock.py): the same states bit for bit, the generator in the same state afterwards -- disc resets (one draw in
    ~140 passes at N = 100), two flocks, leaders with a stricter degree rule, lattice resets (sequential path)."""
    from multiagent_gnn_policies_amd.envs import FlockParams, VecFlock, flocking as fl
    p = FlockParams(**kw)
    op = ofl.FlockParams(**{f: getattr(p, f) for f in ofl.FlockParams.__dataclass_fields__})
    for seed in (0, 7):
        r = np.random.RandomState(seed)
        seq = np.stack([fl.sample_initial_state(r, p) for _ in range(B)])
        tail = r.random_sample()
        r = np.random.RandomState(seed)
        ora = np.stack([ofl.reset(r, op) for _ in range(B)])
        r = np.random.RandomState(seed)
        bat = fl.sample_initial_states(r, p, B, torch.device('cuda'))
        assert np.array_equal(bat, seq) and np.array_equal(bat, ora) and r.random_sample() == tail
    np.random.seed(5)                                            # the module-level generator (what the training loops pass)
    seq = np.stack([fl.sample_initial_state(np.random, p) for _ in range(B)])
    tail = np.random.random_sample()
    np.random.seed(5)
    sim = VecFlock(B, p, 'cuda')
    sim.reset(np.random)
    assert np.array_equal(sim.x.cpu().numpy(), seq) and np.random.random_sample() == tail


@pytest.mark.parametrize('N', [2, 3, 17, 100, 257, 1000])
def test_reset_check_statistics_equal_numpy(N):
    """mgp_flock_reset_check against the numpy evaluation of the spec's acceptance statistics (envs/flocking.py::_candidate_ok):
    min over agents of |{j != i : r2_ij < R^2}| and min over pairs of r2, with r2 = dx dx + dy dy unfused -- bit-equal, on
    random candidates, candidates with coincident agents (r2 = 0), pairs exactly at the radius, and M = 0."""
    import ctypes
    from multiagent_gnn_policies_amd import _lib, ops
    L = _lib.lib()
    rs = np.random.RandomState(N)
    M = 37
    pos = rs.uniform(-3.0, 3.0, size=(M, N, 2))
    pos[1, N - 1] = pos[1, 0]                                     # coincident agents
    pos[2, 0] = (0.0, 0.0); pos[2, 1] = (0.6, 0.8)                # r2 = 1 up to rounding: the strict inequality decides
    R2 = 1.0
    d = pos[:, :, None, :] - pos[:, None, :, :]
    r2 = d[..., 0] * d[..., 0] + d[..., 1] * d[..., 1]
    idx = np.arange(N)
    r2[:, idx, idx] = np.inf
    deg_ref = (r2 < R2).sum(axis=2).min(axis=1)
    r2_ref = r2.reshape(M, -1).min(axis=1)
    pd = torch.from_numpy(pos).cuda()
    deg = torch.full((M,), -7, device='cuda', dtype=torch.int32)
    r2m = torch.full((M,), float('nan'), device='cuda', dtype=torch.float64)
    _lib.check(L.mgp_flock_reset_check(pd.data_ptr(), M, N, ctypes.c_double(R2), deg.data_ptr(), r2m.data_ptr(), ops._stream()),
               'mgp_flock_reset_check')
    assert np.array_equal(deg.cpu().numpy(), deg_ref) and np.array_equal(r2m.cpu().numpy(), r2_ref)
    assert r2_ref[1] == 0.0
    assert L.mgp_flock_reset_check(pd.data_ptr(), 0, N, ctypes.c_double(R2), deg.data_ptr(), r2m.data_ptr(), ops._stream()) == 0
    assert L.mgp_flock_reset_check(pd.data_ptr(), M, 1, ctypes.c_double(R2), deg.data_ptr(), r2m.data_ptr(), ops._stream()) != 0


def test_resets_drawn_under_the_updates_leave_the_run_unchanged(monkeypatch):
    """train_dagger_vec draws the next round's reset states while the GPU runs the current round's updates (disc resets).  The
    same run with MGP_PREFETCH_RESETS=0 -- every round draws its resets when it starts -- must end with the same weights bit
    for bit, the same test statistics, and numpy's / Python's generators in the same state."""
    import random
    from multiagent_gnn_policies_amd.learner.vec_dagger import train_dagger_vec
    cp = configparser.ConfigParser()
    cp['DEFAULT'] = dict(alg='dagger_vec', batch_size='20', buffer_size='2000', updates_per_step='3', seed='1', actor_lr='1e-3',
                         n_train_episodes='24', beta_coeff='0.993', test_interval='40', n_test_episodes='4', k='3',
                         hidden_size='32', gamma='0.99', tau='0.5', env='FlockingRelative-v0', v_max='3.0', comm_radius='1.0',
                         n_agents='60', n_actions='2', n_states='6', debug='False', dt='0.01', init_mode='disc')
    cp['t'] = {}
    outs = []
    for flag in ('1', '0'):
        monkeypatch.setenv('MGP_PREFETCH_RESETS', flag)
        random.seed(2); np.random.seed(2); torch.manual_seed(2)
        st = train_dagger_vec(cp['t'], 'cuda:0', n_envs=8, episode_steps=30)
        outs.append((st['learner'].actor_optim.flat.clone(), st['mean'], st['std'], st['updates'], np.random.random_sample(),
                     random.random()))
    assert outs[0][3] == 3 * 3 * 8 and torch.equal(outs[0][0], outs[1][0]) and outs[0][1:] == outs[1][1:]


def test_train_py_jobs_runs_sections_side_by_side_with_the_sequential_output():
    """`python train.py <cfg> --jobs 3`: the three sections of cfg/smoke_vec_sweep.cfg as three worker processes on the one GPU --
    the printed lines (header, then `section, mean, std` in file order) are those of the sequential run, digit for digit
    (every section seeds its own streams: reference train.py:24-28)."""
    env = dict(os.environ, PYTHONPATH=ROOT)
    outs = []
    for extra in ([], ['--jobs', '3']):
        r = subprocess.run([sys.executable, os.path.join(ROOT, 'train.py'), 'cfg/smoke_vec_sweep.cfg'] + extra, cwd=ROOT, env=env,
                           stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
        assert r.returncode == 0, r.stderr[-3000:]
        outs.append([l for l in r.stdout.strip().splitlines() if l.strip()])
    assert outs[0] == outs[1], outs
    assert outs[0][0] == 'alg, reward' and [l.split(',')[0] for l in outs[0][1:]] == ['k3', 'k2', 'n40']
