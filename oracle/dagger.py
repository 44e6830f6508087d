"""numpy restatement of the DAGGER update arithmetic (TEST ORACLE).

Follows reference learner/gnn_dagger.py:
  * select_action: out (1,1,nA,N) -> action (N,nA) = out[0,0].T        :66-68
  * expert label (N,nA) -> (1,1,nA,N)                                  :174-176
  * loss = mean over (B,1,nA,N) of (pred - label)^2  (F.mse_loss)      :91
  * Adam(lr), torch defaults betas=(0.9,0.999), eps=1e-8, no decay     :49,93
"""
import numpy as np

from . import actor as _actor


def action_from_output(out):
    """gnn_dagger.py:66-68 : (1,1,nA,N) -> (N,nA)"""
    out = np.asarray(out)
    assert out.shape[0] == 1 and out.shape[1] == 1
    return np.ascontiguousarray(out[0, 0].T)


def label_from_action(action):
    """gnn_dagger.py:174-176 : (N,nA) -> (1,1,nA,N)"""
    a = np.asarray(action)
    return np.ascontiguousarray(a.T).reshape(1, 1, a.shape[1], a.shape[0])


def mse_loss(pred, target, dtype=np.float64):
    d = np.asarray(pred, dtype=dtype) - np.asarray(target, dtype=dtype)
    return float(np.mean(d * d))


def mse_grad(pred, target, dtype=np.float64):
    """d loss / d pred for loss = mean((pred-target)^2)."""
    d = np.asarray(pred, dtype=dtype) - np.asarray(target, dtype=dtype)
    return (2.0 / d.size) * d


def adam_step(params, grads, m, v, t, lr, beta1=0.9, beta2=0.999, eps=1e-8, dtype=np.float32):
    """torch.optim.Adam single step (no amsgrad / weight decay), t is the 1-based step count.

    p <- p - lr/(1-b1^t) * m / (sqrt(v)/sqrt(1-b2^t) + eps)
    """
    out_p, out_m, out_v = [], [], []
    bc1 = 1.0 - beta1 ** t
    bc2 = 1.0 - beta2 ** t
    for p, g, mi, vi in zip(params, grads, m, v):
        p = np.asarray(p, dtype=dtype)
        g = np.asarray(g, dtype=dtype)
        mi = (beta1 * np.asarray(mi, dtype=dtype) + (1.0 - beta1) * g).astype(dtype)
        vi = (beta2 * np.asarray(vi, dtype=dtype) + (1.0 - beta2) * g * g).astype(dtype)
        denom = np.sqrt(vi) / np.sqrt(bc2) + eps
        p = (p - (lr / bc1) * (mi / denom)).astype(dtype)
        out_p.append(p)
        out_m.append(mi)
        out_v.append(vi)
    return out_p, out_m, out_v


def gradient_step(delay_state, delay_gso, labels, weights, biases, ind_agg, m, v, t, lr,
                  dtype=np.float32):
    """One DAGGER.gradient_step (gnn_dagger.py:83-96) on already-concatenated batches.

    Returns (loss, new_weights, new_biases, new_m, new_v, grads) with params ordered
    [W0,b0,W1,b1,...] in m / v / grads, like actor.parameters().
    """
    out, cache = _actor.forward(delay_state, delay_gso, weights, biases, ind_agg,
                                dtype=dtype, return_cache=True)
    loss = mse_loss(out, labels)
    d_out = mse_grad(out, labels, dtype=dtype)
    dWs, dbs, _ = _actor.backward(d_out, delay_gso, weights, ind_agg, cache, dtype=dtype)
    params, grads = [], []
    for W, b, dW, db in zip(weights, biases, dWs, dbs):
        params += [W, b]
        grads += [dW.astype(dtype), db.astype(dtype)]
    new_p, new_m, new_v = adam_step(params, grads, m, v, t, lr, dtype=dtype)
    return loss, new_p[0::2], new_p[1::2], new_m, new_v, grads
