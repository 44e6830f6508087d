"""numpy restatement of the Actor policy network (TEST ORACLE, not product code).

Follows reference learner/actor.py:
  * layer construction            actor.py:23-42   (layers = [n_s]+hidden+[n_a]; layer `ind_agg`
                                                    has kernel/stride (k,1), the others (1,1))
  * forward                       actor.py:63-82
  * aggregation                   actor.py:69-71   y[b,k,c,n] = sum_m x[b,k,c,m] * G[b,k,m,n]
  * conv + tanh (not on last)     actor.py:73-77
  * final view (B,1,nA,N)         actor.py:82

Weights use the reference's state_dict layout: W_i has shape (out, in, step, 1) with
step = k if i == ind_agg else 1; b_i has shape (out,).

The internal activation layout is the reference's permuted one, T[b, c, k, n].
"""
import numpy as np


def layer_dims(n_s, n_a, hidden_layers):
    """actor.py:23-24"""
    layers = [n_s] + list(hidden_layers) + [n_a]
    return layers, len(layers) - 1


def weight_shapes(n_s, n_a, hidden_layers, k, ind_agg):
    """Shapes of conv_layers.{i}.weight / .bias (actor.py:30-38)."""
    layers, n_layers = layer_dims(n_s, n_a, hidden_layers)
    shapes = []
    for i in range(n_layers):
        step = k if i == ind_agg else 1
        shapes.append(((layers[i + 1], layers[i], step, 1), (layers[i + 1],)))
    return shapes


def aggregate(T, G):
    """T: (B,C,K,N) ; G: (B,K,N,N) -> (B,C,K,N).   actor.py:69-71

    out[b,c,k,n] = sum_m T[b,c,k,m] * G[b,k,m,n]  (contraction over the ROW index of G).
    """
    return np.einsum('bckm,bkmn->bckn', T, G)


def aggregate_bkfn(X, G):
    """Same contraction in the input layout: X (B,K,F,N), G (B,K,N,N) -> (B,K,F,N)."""
    return np.einsum('bkfm,bkmn->bkfn', X, G)


def forward(delay_state, delay_gso, weights, biases, ind_agg, dtype=np.float32, return_cache=False):
    """Actor.forward.  delay_state (B,K,F,N), delay_gso (B,K,N,N) -> (B,1,nA,N).

    `weights[i]` (out,in,step,1), `biases[i]` (out,).
    With return_cache=True also returns the per-layer inputs/outputs needed by backward().
    """
    X = np.asarray(delay_state, dtype=dtype)
    G = np.asarray(delay_gso, dtype=dtype)
    B, K, F, N = X.shape
    assert G.shape == (B, K, N, N)                      # actor.py:55-57,61
    n_layers = len(weights)
    T = np.transpose(X, (0, 2, 1, 3))                   # (B,F,K,N)   actor.py:64
    cache = {'layer_in': [], 'layer_out': [], 'pre_agg': None}
    for i in range(n_layers):
        W = np.asarray(weights[i], dtype=dtype)
        b = np.asarray(biases[i], dtype=dtype)
        if i == ind_agg:
            cache['pre_agg'] = T
            T = aggregate(T, G)                         # actor.py:68-71
        cache['layer_in'].append(T)
        step = W.shape[2]
        if step == 1:
            # 1x1 conv over channels, applied per remaining tap
            Z = np.einsum('oc,bckn->bokn', W[:, :, 0, 0], T) + b[None, :, None, None]
        else:
            # (k,1) kernel with stride (k,1): contracts channels and taps jointly
            assert T.shape[2] == step, "tap axis must equal the kernel height"
            Z = np.einsum('ock,bckn->bon', W[:, :, :, 0], T)[:, :, None, :] + b[None, :, None, None]
        if i < n_layers - 1:
            Z = np.tanh(Z)                              # actor.py:75-77
        cache['layer_out'].append(Z)
        T = Z
    n_a = T.shape[1]
    assert T.shape[2] == 1, "view(B,1,nA,N) requires the tap axis to have collapsed"
    out = T.reshape(B, 1, n_a, N)                       # actor.py:82
    if return_cache:
        return out, cache
    return out


def backward(d_out, delay_gso, weights, ind_agg, cache, dtype=np.float64, need_dx=False):
    """Gradients of sum(d_out * forward(...)) w.r.t. weights/biases (and delay_state).

    Restates what autograd derives for actor.py:63-82:
      delta_l   = dZ_l * (1 - Z_l^2)           (tanh layers)
      dW_l      = sum_{b,(k),n} delta_l (x) input_l ; db_l = sum delta_l
      dInput_l  = W_l^T delta_l
      through the aggregation: dT[b,c,k,m] = sum_n dY[b,c,k,n] * G[b,k,m,n]
    """
    G = np.asarray(delay_gso, dtype=dtype)
    n_layers = len(weights)
    B = d_out.shape[0]
    N = d_out.shape[3]
    dZ = np.asarray(d_out, dtype=dtype).reshape(B, -1, 1, N)   # (B,nA,1,N)
    dWs = [None] * n_layers
    dbs = [None] * n_layers
    for i in reversed(range(n_layers)):
        W = np.asarray(weights[i], dtype=dtype)
        Z = np.asarray(cache['layer_out'][i], dtype=dtype)
        Tin = np.asarray(cache['layer_in'][i], dtype=dtype)
        if i < n_layers - 1:
            delta = dZ * (1.0 - Z * Z)
        else:
            delta = dZ
        step = W.shape[2]
        dbs[i] = delta.sum(axis=(0, 2, 3))
        if step == 1:
            dWs[i] = np.einsum('bokn,bckn->oc', delta, Tin)[:, :, None, None]
            dT = np.einsum('oc,bokn->bckn', W[:, :, 0, 0], delta)
        else:
            dWs[i] = np.einsum('bon,bckn->ock', delta[:, :, 0, :], Tin)[:, :, :, None]
            dT = np.einsum('ock,bon->bckn', W[:, :, :, 0], delta[:, :, 0, :])
        if i == ind_agg:
            dT = np.einsum('bckn,bkmn->bckm', dT, G)
        dZ = dT
    dX = np.transpose(dZ, (0, 2, 1, 3)) if need_dx else None   # back to (B,K,F,N)
    return dWs, dbs, dX
