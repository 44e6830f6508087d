#!/usr/bin/env python3
"""Generate the golden fixtures in tests/golden/ by RUNNING THE REFERENCE ITSELF.

Run in the build container only (needs /root/reference):

    python tests/golden/gen_golden.py

It imports learner.actor.Actor, learner.state_with_delay.MultiAgentStateWithDelay and
learner.gnn_dagger.DAGGER from /root/reference, feeds them seeded synthetic inputs from
oracle/synth.py, and stores inputs' checksums + the reference's outputs as small .npz
files.  Only data (arrays) is written -- no reference source, bytecode or pickles.
The shipped checkpoint's six tensors are stored as plain arrays (they are data).
"""
import configparser
import os
import sys
from types import SimpleNamespace

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = '/root/reference'
sys.path.insert(0, ROOT)
sys.path.insert(0, REF)

from oracle import synth  # noqa: E402
from learner.actor import Actor as RefActor  # noqa: E402
from learner.state_with_delay import MultiAgentStateWithDelay as RefState  # noqa: E402
from learner.gnn_dagger import DAGGER as RefDAGGER  # noqa: E402
from learner.replay_buffer import Transition as RefTransition  # noqa: E402

torch.set_num_threads(1)


def sd_to_np(sd):
    return {k.replace('.', '__'): v.detach().cpu().numpy().copy() for k, v in sd.items()}


def make_args(**kw):
    cp = configparser.ConfigParser()
    base = dict(alg='dagger', batch_size='20', buffer_size='10000', updates_per_step='200', seed='11',
                actor_lr='5e-5', n_train_episodes='400', beta_coeff='0.993', test_interval='40',
                n_test_episodes='20', k='3', hidden_size='32', gamma='0.99', tau='0.5',
                env='FlockingRelative-v0', v_max='3.0', comm_radius='1.0', n_agents='100',
                n_actions='2', n_states='6', debug='False', dt='0.01')
    base.update({k: str(v) for k, v in kw.items()})
    cp['DEFAULT'] = base
    cp['test'] = {}
    return cp['test']


def actor_case(name, actor, seed, B, K, F, N, ind_agg, dense=False, store_inputs=False):
    X, G = (synth.make_dense_inputs if dense else synth.make_inputs)(seed, B, K, F, N)
    xt = torch.from_numpy(X).requires_grad_(True)
    gt = torch.from_numpy(G)
    out = actor(xt, gt)
    rs = np.random.RandomState(seed + 101)
    target = rs.randn(*out.shape).astype(np.float32)
    loss = torch.nn.functional.mse_loss(out, torch.from_numpy(target))
    actor.zero_grad()
    loss.backward()
    d = dict(seed=seed, shape=np.array([B, K, F, N]), ind_agg=ind_agg, dense=int(dense),
             in_checksum=synth.checksum(X, G), out=out.detach().numpy(), target=target,
             loss=np.float64(loss.item()), dX=xt.grad.numpy())
    hidden = [l.out_channels for l in actor.conv_layers][:-1]
    d['hidden'] = np.array(hidden, dtype=np.int64)
    for k, v in sd_to_np(actor.state_dict()).items():
        d['w__' + k] = v
    for n, p in actor.named_parameters():
        d['g__' + n.replace('.', '__')] = p.grad.detach().numpy().copy()
    if store_inputs:
        d['X'] = X
        d['G'] = G
    path = os.path.join(HERE, name + '.npz')
    np.savez_compressed(path, **d)
    print('wrote', path, 'out absmax', float(np.abs(d['out']).max()), 'loss', float(d['loss']))


def gen_actor():
    # shipped checkpoint (K=3, H=32, n_layers=2, ind_agg=0)
    ck = torch.load(os.path.join(REF, 'models', 'actor_FlockingRelative-v0_dagger_k3'), map_location='cpu')
    np.savez_compressed(os.path.join(HERE, 'ckpt_dagger_k3.npz'), **sd_to_np(ck))
    a = RefActor(6, 2, [32, 32], 3, 0)
    a.load_state_dict(ck)
    actor_case('actor_ckpt_B1_N100', a, 0, 1, 3, 6, 100, 0)
    actor_case('actor_ckpt_B4_N100', a, 1, 4, 3, 6, 100, 0)
    actor_case('actor_ckpt_B3_N100_dense', a, 2, 3, 3, 6, 100, 0, dense=True)
    actor_case('actor_ckpt_B2_N16', a, 3, 2, 3, 6, 16, 0, store_inputs=True)
    # default-init variants (torch.manual_seed(s) -> Conv2d default init)
    variants = [
        # name, seed, B, K, F, N, hidden, ind_agg
        ('actor_init_K4_N200', 0, 2, 4, 6, 200, [32, 32], 0),
        ('actor_init_K2_N16', 1, 2, 2, 6, 16, [32, 32], 0),
        ('actor_init_K1_N16', 2, 2, 1, 6, 16, [32, 32], 0),
        ('actor_init_L1_H8_N16', 3, 2, 3, 6, 16, [8], 0),
        ('actor_init_L1_H4_N100', 4, 1, 3, 6, 100, [4], 0),
        ('actor_init_H128_N33', 5, 2, 3, 6, 33, [128, 128], 0),
        ('actor_init_L3_H16_N16', 6, 2, 3, 6, 16, [16, 16, 16], 0),
        ('actor_init_L4_H64_N50', 7, 3, 3, 6, 50, [64, 64, 64, 64], 0),
        ('actor_init_agg1_N16', 8, 2, 3, 6, 16, [16, 16], 1),
        ('actor_init_agg2_N20', 9, 2, 2, 6, 20, [16, 16, 16], 2),
        ('actor_init_agg1_N100', 10, 2, 3, 6, 100, [32, 32], 1),
        ('actor_init_F3_A1_N7', 11, 3, 2, 3, 7, [5], 0),
        ('actor_init_nohidden_N16', 12, 2, 3, 6, 16, [], 0),
    ]
    for name, seed, B, K, F, N, hidden, ia in variants:
        torch.manual_seed(seed)
        n_a = 1 if 'A1' in name else 2
        a = RefActor(F, n_a, hidden, K, ia)
        actor_case(name, a, seed, B, K, F, N, ia, store_inputs=(N <= 20))


def env_tuple(rs, n, f):
    """A fake env observation: (values (n,f) f64, network (n,n) f64 row-normalised, zero diag)."""
    vals = rs.randn(n, f)
    net = synth.geometric_adjacency(rs, n)
    return vals, net


def gen_state():
    for name, n, k, steps, full in [('state_N16_K3', 16, 3, 5, True), ('state_N16_K1', 16, 1, 3, True),
                                    ('state_N16_K4', 16, 4, 6, True), ('state_N100_K3', 100, 3, 5, False)]:
        args = make_args(n_agents=n, k=k)
        rs = np.random.RandomState(1234 + n + k)
        prev = None
        d = dict(n=n, k=k, steps=steps)
        for t in range(steps):
            vals, net = env_tuple(rs, n, 6)
            st = RefState(torch.device('cpu'), args, (vals, net), prev_state=prev)
            if full or t == steps - 1:
                d[f'delay_gso_{t}'] = st.delay_gso.numpy()
                d[f'delay_state_{t}'] = st.delay_state.numpy()
                d[f'curr_gso_{t}'] = st.curr_gso.numpy()
            d[f'values_{t}'] = vals
            if full:
                d[f'network_{t}'] = net
            d[f'cs_{t}'] = synth.checksum(st.delay_gso.numpy(), st.delay_state.numpy(), st.curr_gso.numpy())
            prev = st
        path = os.path.join(HERE, name + '.npz')
        np.savez_compressed(path, **d)
        print('wrote', path)


def gen_dagger():
    """DAGGER.select_action and three consecutive gradient_steps (loss + post-Adam weights)."""
    for name, n, k, bsz in [('dagger_N16_K3', 16, 3, 20), ('dagger_N100_K3', 100, 3, 20)]:
        args = make_args(n_agents=n, k=k, batch_size=bsz)
        torch.manual_seed(11)
        learner = RefDAGGER(torch.device('cpu'), args)
        d = dict(n=n, k=k, bsz=bsz, lr=5e-5)
        for kk, v in sd_to_np(learner.actor.state_dict()).items():
            d['w0__' + kk] = v
        # select_action on a B=1 state
        X1, G1 = synth.make_inputs(77, 1, k, 6, n)
        st = SimpleNamespace(delay_state=torch.from_numpy(X1), delay_gso=torch.from_numpy(G1))
        act = learner.select_action(st)
        d['select_action'] = act.numpy().copy()
        d['select_cs'] = synth.checksum(X1, G1)
        losses = []
        for step in range(3):
            X, G = synth.make_inputs(200 + step, bsz, k, 6, n)
            rs = np.random.RandomState(300 + step)
            labels = rs.randn(bsz, 1, 2, n).astype(np.float32)
            states = [SimpleNamespace(delay_state=torch.from_numpy(X[i:i + 1]),
                                      delay_gso=torch.from_numpy(G[i:i + 1])) for i in range(bsz)]
            actions = [torch.from_numpy(labels[i:i + 1]) for i in range(bsz)]
            batch = RefTransition(tuple(states), tuple(actions), None, None, None)
            losses.append(learner.gradient_step(batch))
            for kk, v in sd_to_np(learner.actor.state_dict()).items():
                d[f'w{step + 1}__' + kk] = v.copy()
            d[f'in_cs_{step}'] = synth.checksum(X, G, labels)
        d['losses'] = np.array(losses, dtype=np.float64)
        path = os.path.join(HERE, name + '.npz')
        np.savez_compressed(path, **d)
        print('wrote', path, 'losses', losses)


def gen_train_trace():
    """Row a9: the reference's OWN `train_dagger` (gnn_dagger.py:126-243) and ReplayBuffer (replay_buffer.py:6-49) driven
    by a deterministic duck-typed environment (tests/fake_env.py); tests/trace_tools.py records the control flow:
    beta passed to every coin flip and its outcome, who drove each step, every action handed to env.step, labels and ring
    positions of every insert, the indices of every minibatch, per-update losses, select_action outputs, printed lines,
    final statistics and weights, and the order of all these events."""
    import learner.gnn_dagger as ref_mod
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    import fake_env
    import trace_tools as tt
    if not hasattr(np, 'Inf'):
        np.Inf = np.inf        # gnn_dagger.py:143 spells it np.Inf (removed in numpy 2.0); a shim for RUNNING the reference here
    args = tt.trace_args()
    tr = tt.Trace()
    env = tt.RecordingEnv(fake_env.FakeFlockEnv(args.getint('n_agents'), episode_steps=tt.TRACE_EPISODE_STEPS,
                                                seed=args.getint('seed')), tr)
    saved = (ref_mod.ReplayBuffer, ref_mod.DAGGER)
    ref_mod.ReplayBuffer = tt.recording_replay(saved[0], tr)
    ref_mod.DAGGER = tt.recording_learner(saved[1], tr, lambda l: sd_to_np(l.actor.state_dict()))
    try:
        tt.seed_all(args.getint('seed'))
        with tt.recording_binomial(tr), tt.capture_stdout(tr):
            tr.stats = ref_mod.train_dagger(env, args, torch.device('cpu'))
    finally:
        ref_mod.ReplayBuffer, ref_mod.DAGGER = saved
    tr.final_weights = sd_to_np(tr.learner.actor.state_dict())
    d = tr.to_npz_dict()
    d['episode_steps'] = np.int64(tt.TRACE_EPISODE_STEPS)
    for k, v in tt.TRACE_CFG.items():
        d['cfg__' + k] = np.array(v)
    path = os.path.join(HERE, 'train_dagger_trace.npz')
    np.savez_compressed(path, **d)
    print('wrote', path)
    print(' events', ''.join(tr.events))
    print(' printed:\n' + tr.printed)
    print(' stats', tr.stats, 'losses', tr.losses[:4], '...', len(tr.losses), 'updates;', len(tr.step_actions), 'env steps;',
          'expert drove', int(np.sum(tr.step_expert_applied)), 'of', len(tr.binom_out), 'training steps')


if __name__ == '__main__':
    which = sys.argv[1:] or ['actor', 'state', 'dagger', 'trace']
    if 'actor' in which:
        gen_actor()
    if 'state' in which:
        gen_state()
    if 'dagger' in which:
        gen_dagger()
    if 'trace' in which:
        gen_train_trace()
