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
