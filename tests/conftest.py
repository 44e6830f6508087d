import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_sessionstart(session):
    """The allowance log (check_parity below) describes ONE run of the GPU suite."""
    try:
        import torch
        if torch.cuda.is_available():
            os.remove(os.path.join(ROOT, 'gpurun_out', 'parity_allowances.jsonl'))
    except (OSError, ImportError):
        pass


def load_golden(name):
    path = os.path.join(GOLDEN, name if name.endswith('.npz') else name + '.npz')
    with np.load(path) as z:
        return {k: z[k] for k in z.files}


def golden_weights(g, prefix='w__'):
    """Extract ([W_i], [b_i]) from a golden dict holding `<prefix>conv_layers__i__weight/bias`."""
    Ws, bs = [], []
    i = 0
    while f'{prefix}conv_layers__{i}__weight' in g:
        Ws.append(g[f'{prefix}conv_layers__{i}__weight'])
        bs.append(g[f'{prefix}conv_layers__{i}__bias'])
        i += 1
    return Ws, bs


def golden_grads(g):
    return golden_weights(g, prefix='g__')


ACTOR_GOLDENS = sorted(f[:-4] for f in os.listdir(GOLDEN) if f.startswith('actor_') and f.endswith('.npz'))
STATE_GOLDENS = sorted(f[:-4] for f in os.listdir(GOLDEN) if f.startswith('state_') and f.endswith('.npz'))
DAGGER_GOLDENS = sorted(f[:-4] for f in os.listdir(GOLDEN) if f.startswith('dagger_') and f.endswith('.npz'))


def golden_inputs(g):
    """Regenerate (X, G) of an actor golden from its seed and verify the stored checksum."""
    from oracle import synth
    B, K, F, N = [int(v) for v in g['shape']]
    seed = int(g['seed'])
    if 'X' in g:
        X, G = g['X'], g['G']
    else:
        X, G = (synth.make_dense_inputs if int(g['dense']) else synth.make_inputs)(seed, B, K, F, N)
    assert abs(synth.checksum(X, G) - float(g['in_checksum'])) <= 1e-9 * max(1.0, abs(float(g['in_checksum']))), \
        "synthetic input generator drifted from the one the goldens were made with"
    return X, G


@pytest.fixture(scope='session')
def has_gpu():
    import torch
    return torch.cuda.is_available()


# ------------------------------------------------------------------------------------------------------------------------
# The allowance around north_star's 1e-5 (elementwise, against the fp64 evaluation `ref` of the reference op sequence on the
# identical fp32 (S, X)):   |gpu - ref| <= 1e-5 * max(1, |ref|)                       wherever the tests can afford it, else
#                           |gpu - ref| <= (1e-5 + NOISE_FACTOR * noise) * max(1, |ref|)
# where `noise` is how far fp32 evaluations of the REFERENCE itself are from `ref` on these very inputs (colliding agents make
# 1/r^4 features of 1e4 and more; x3 random weights and sum pooling amplify them): "within 1e-5 of the fp32 reference" can
# only be asked to within the reference's own fp32 scatter.  noise = the LARGEST of three witnesses (reference_noise);
# NOISE_FACTOR = 2: the kernel's error is another draw from the distribution the witnesses sample (round 3 used 10 on a
# single witness).  Every check is logged (gpurun_out/parity_allowances.jsonl) with the factor it would have needed.
NOISE_FACTOR = float(os.environ.get('MGP_NOISE_FACTOR', '2.0'))
_ALLOWANCE_LOG = os.path.join(ROOT, 'gpurun_out', 'parity_allowances.jsonl')


def reference_noise(X, G, Ws, bs, K=None, per_episode=False):
    """Distance of fp32 evaluations of the reference Actor forward from its fp64 evaluation on (X, G) -- the largest of:
    the op sequence in numpy fp32 (oracle/actor.py), in PyTorch-CPU fp32 (oracle/torch_port.py: the reference's own
    framework and op order, actor.py:63-82), and the fp64 evaluation of inputs moved by ONE fp32 rounding (every element
    times (1 +- 2^-24), fixed seed).  Elementwise |.| / max(1, |ref|), maximum over all elements (or per episode)."""
    import torch
    from oracle import actor as oa, torch_port
    X = np.asarray(X); G = np.asarray(G)
    ref = oa.forward(X.astype(np.float64), G.astype(np.float64), Ws, bs, 0, dtype=np.float64)
    a = oa.forward(X.astype(np.float32), G.astype(np.float32), Ws, bs, 0, dtype=np.float32)
    with torch.no_grad():
        b = torch_port.actor_forward(torch.from_numpy(X.astype(np.float32)), torch.from_numpy(G.astype(np.float32)),
                                     [torch.from_numpy(np.asarray(w, dtype=np.float32)) for w in Ws],
                                     [torch.from_numpy(np.asarray(v, dtype=np.float32)) for v in bs], 0,
                                     K if K is not None else X.shape[1]).numpy()
    rs = np.random.RandomState(12345)
    eps = 2.0 ** -24
    Xp = X.astype(np.float64) * (1.0 + eps * rs.choice([-1.0, 1.0], size=X.shape))
    Gp = G.astype(np.float64) * (1.0 + eps * rs.choice([-1.0, 1.0], size=G.shape))
    c = oa.forward(Xp, Gp, Ws, bs, 0, dtype=np.float64)
    den = np.maximum(1.0, np.abs(ref))
    e = np.maximum.reduce([np.abs(np.asarray(w_, dtype=np.float64) - ref) / den for w_ in (a, b, c)])
    if per_episode:
        return e.reshape(e.shape[0], -1).max(axis=1), ref
    return float(e.max()), ref


def check_parity(u, ref, noise, what, factor=None):
    """Assert the allowance above elementwise; `noise` scalar or per episode.  Logs err / noise / the factor needed."""
    import json
    factor = NOISE_FACTOR if factor is None else factor
    u = np.asarray(u, dtype=np.float64); ref = np.asarray(ref, dtype=np.float64)
    err = (np.abs(u - ref) / np.maximum(1.0, np.abs(ref))).reshape(u.shape[0], -1).max(axis=1)
    nz = np.broadcast_to(np.asarray(noise, dtype=np.float64), err.shape)
    need = float(np.max(np.where(err > 1e-5, (err - 1e-5) / np.maximum(nz, 1e-300), 0.0)))
    try:
        os.makedirs(os.path.dirname(_ALLOWANCE_LOG), exist_ok=True)
        with open(_ALLOWANCE_LOG, 'a') as f:
            f.write(json.dumps(dict(test=os.environ.get('PYTEST_CURRENT_TEST', ''), what=str(what), err=float(err.max()),
                                    noise=float(nz.max()), factor_needed=need, factor_allowed=factor,
                                    plain=bool(np.all(err <= 1e-5)))) + '\n')
    except OSError:
        pass
    assert np.all(err <= 1e-5 + factor * nz), (what, 'err %.3g' % err.max(), 'noise %.3g' % nz.max(), 'factor needed %.2f' % need)
    return float(err.max())
