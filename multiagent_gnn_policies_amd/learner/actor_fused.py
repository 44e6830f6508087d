"""Fused single-kernel Actor forward/backward (ind_agg == 0) on top of mgp_actor_fwd / mgp_actor_bwd."""
import ctypes

import torch

from .. import _lib, ops


def _ptr_array(tensors):
    arr = (ctypes.c_void_p * len(tensors))()
    for i, t in enumerate(tensors):
        arr[i] = t.data_ptr()
    return arr


class _ActorFusedFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, X, G, dims, K, want_grad, *params):
        L = _lib.lib()
        n_layers = len(dims) - 1
        B, _, F, N = X.shape
        Ws = [p.contiguous() for p in params[0::2]]
        bs = [p.contiguous() for p in params[1::2]]
        cdims = (ctypes.c_int * len(dims))(*dims)
        need_bwd = want_grad and any(ctx.needs_input_grad[5:])
        saved = None
        if need_bwd:
            n_saved = L.mgp_actor_saved_floats(cdims, n_layers, B, K, N)
            saved = torch.empty((n_saved,), device=X.device, dtype=torch.float32)
        out = torch.empty((B, 1, dims[-1], N), device=X.device, dtype=torch.float32)
        rc = L.mgp_actor_fwd(ops._ptr(X), ops._ptr(G), _ptr_array(Ws), _ptr_array(bs), cdims, n_layers,
                             ops._ptr(out), ops._ptr(saved), B, K, N, ops._stream())
        _lib.check(rc, 'mgp_actor_fwd')
        ctx.dims, ctx.K, ctx.shape = dims, K, (B, F, N)
        ctx.save_for_backward(saved, *Ws)
        ctx.param_shapes = [p.shape for p in params]
        return out

    @staticmethod
    def backward(ctx, dOut):
        if ctx.needs_input_grad[0] or ctx.needs_input_grad[1]:
            raise ops.MgpError("fused Actor backward provides parameter gradients only")
        L = _lib.lib()
        saved, *Ws = ctx.saved_tensors
        dims, K = ctx.dims, ctx.K
        B, F, N = ctx.shape
        n_layers = len(dims) - 1
        cdims = (ctypes.c_int * len(dims))(*dims)
        dWs = [torch.empty_like(w) for w in Ws]
        dbs = [torch.empty((dims[i + 1],), device=dOut.device, dtype=torch.float32) for i in range(n_layers)]
        n_ws = L.mgp_actor_bwd_workspace(cdims, n_layers, B, K, N)
        ws = torch.empty((max(1, n_ws),), device=dOut.device, dtype=torch.float32)
        rc = L.mgp_actor_bwd(ops._ptr(dOut.contiguous()), ops._ptr(saved), _ptr_array(Ws), cdims, n_layers,
                             _ptr_array(dWs), _ptr_array(dbs), B, K, N, ops._ptr(ws), ops._stream())
        _lib.check(rc, 'mgp_actor_bwd')
        grads = []
        for i in range(n_layers):
            grads += [dWs[i].view(ctx.param_shapes[2 * i]), dbs[i]]
        return (None, None, None, None, None) + tuple(grads)


def _try_forward_deep(actor, delay_state, delay_gso, cdims):
    """Inference with three or more hidden layers beyond mgp_actor_fwd's LDS plan (hidden_size 128 at n_layers 3, 4):
    mgp_actor_fwd_deep, one launch.  None when not covered (or gradients are wanted: the composed ops carry autograd)."""
    if torch.is_grad_enabled() and any(p.requires_grad for p in actor.parameters()):
        return None
    L = _lib.lib()
    B, K, _, N = delay_state.shape
    if not L.mgp_actor_deep_supported(cdims, actor.n_layers, actor.k, N):
        return None
    X, G = delay_state.contiguous(), delay_gso.contiguous()
    Ws = [c.weight.view(c.weight.shape[0], -1).contiguous() for c in actor.conv_layers]
    bs = [c.bias.contiguous() for c in actor.conv_layers]
    if (X.data_ptr() | G.data_ptr()) & 15 or any(w.data_ptr() & 15 for w in Ws[1:-1]):
        return None
    out = torch.empty((B, 1, actor.layers[-1], N), device=X.device, dtype=torch.float32)
    rc = L.mgp_actor_fwd_deep(ops._ptr(X), ops._ptr(G), _ptr_array(Ws), _ptr_array(bs), cdims, actor.n_layers,
                              ops._ptr(out), B, actor.k, N, ops._stream())
    _lib.check(rc, 'mgp_actor_fwd_deep')
    return out


def try_forward(actor, delay_state, delay_gso):
    """Returns the (B,1,nA,N) output, or None when the fused kernel does not cover this shape
    (the caller then composes the generic HIP ops)."""
    if delay_state.requires_grad or delay_gso.requires_grad:
        return None
    if actor.n_layers > 8:
        return None
    ops._dev(delay_state, 'delay_state'); ops._dev(delay_gso, 'delay_gso')
    dims = tuple(actor.layers)
    cdims = (ctypes.c_int * len(dims))(*dims)
    if not _lib.lib().mgp_actor_supported(cdims, actor.n_layers, actor.k, delay_state.shape[3]):
        return _try_forward_deep(actor, delay_state, delay_gso, cdims)
    X = delay_state.contiguous()
    G = delay_gso.contiguous()
    if max(dims[1:]) > 64 and ((X.data_ptr() | G.data_ptr()) & 15):
        return None         # widths above 64 are covered by the MFMA-aggregation variant only, which loads aligned quads
    params = []
    for conv in actor.conv_layers:
        out_c = conv.weight.shape[0]
        params += [conv.weight.view(out_c, -1), conv.bias]
    # under torch.no_grad() (select_action / rollouts) nothing is saved for backward: Y and Z stay on chip
    return _ActorFusedFn.apply(X, G, tuple(actor.layers), actor.k, torch.is_grad_enabled(), *params)
