"""torch-tensor front end of the C ABI (include/mgp.h): argument checking, raw pointers, the current
HIP stream, and the autograd glue.  Every function here runs a hand-written HIP kernel from libmgp.so;
nothing falls back to ATen math or to the CPU.
"""
import contextlib
import ctypes

import torch

from . import _lib
from ._lib import MgpError, MgpFlockParams

ACT_NONE, ACT_TANH = 0, 1


_raw_stream = getattr(torch._C, '_cuda_getCurrentRawStream', None)


def _stream():
    """Handle of the HIP stream torch is currently enqueuing on (honours torch.cuda.stream(...) and graph capture).
    The raw accessor costs ~1 us; building a torch.cuda.Stream object first costs ~10 us -- four of them per
    environment step of the one-environment loops."""
    if _raw_stream is not None:
        return ctypes.c_void_p(_raw_stream(torch.cuda.current_device()))
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _dev(t, name, dtype=torch.float32):
    if not isinstance(t, torch.Tensor):
        raise TypeError("%s must be a torch.Tensor" % name)
    if not t.is_cuda:
        raise MgpError("%s is on %s: the MI355X kernels need tensors on a HIP device (there is no CPU "
                       "fallback in this package)" % (name, t.device))
    if t.dtype != dtype:
        raise MgpError("%s must be %s, got %s" % (name, dtype, t.dtype))
    return t


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)


@contextlib.contextmanager
def graph_capture(graph):
    """`torch.cuda.graph(graph)` with Python's cyclic collector held off.  A collection DURING capture can run the destructor
    of an unrelated device object (an older CUDAGraph, an event of an earlier test or training round); its HIP call is illegal
    on a capturing thread and the runtime aborts the process -- seen once in three full GPU test runs ("Fatal Python error:
    Aborted ... Garbage-collecting" under an update's _enqueue).  The collector stays off until the capture ends."""
    import gc
    was = gc.isenabled()
    gc.disable()                          # (torch.cuda.graph collects once on entry by itself)
    try:
        with torch.cuda.graph(graph):
            yield
    finally:
        if was:
            gc.enable()


# ------------------------------------------------------------------------------------ aggregation
def agg_fwd(T, G):
    """T (B,C,K,N) (any b/c/k strides, agent axis contiguous), G (B,K,N,N) -> (B,C,K,N) contiguous.
    out[b,c,k,n] = sum_m T[b,c,k,m] G[b,k,m,n]          (reference actor.py:69-71)"""
    _dev(T, 'T'); _dev(G, 'G')
    B, C, K, N = T.shape
    assert G.shape == (B, K, N, N)
    if not G.is_contiguous():
        G = G.contiguous()
    if T.stride(3) != 1:
        T = T.contiguous()
    Y = torch.empty((B, C, K, N), device=T.device, dtype=torch.float32)
    rc = _lib.lib().mgp_agg_fwd(_ptr(T), _ptr(G), _ptr(Y), B, K, C, N,
                                T.stride(0), T.stride(2), T.stride(1),
                                Y.stride(0), Y.stride(2), Y.stride(1), _stream())
    _lib.check(rc, 'mgp_agg_fwd')
    return Y


def agg_bwd_x(dY, G):
    """dY (B,C,K,N), G (B,K,N,N) -> dT (B,C,K,N): dT[b,c,k,m] = sum_n dY[b,c,k,n] G[b,k,m,n]."""
    _dev(dY, 'dY'); _dev(G, 'G')
    B, C, K, N = dY.shape
    if not G.is_contiguous():
        G = G.contiguous()
    if dY.stride(3) != 1:
        dY = dY.contiguous()
    dT = torch.empty((B, C, K, N), device=dY.device, dtype=torch.float32)
    rc = _lib.lib().mgp_agg_bwd_x(_ptr(dY), _ptr(G), _ptr(dT), B, K, C, N,
                                  dY.stride(0), dY.stride(2), dY.stride(1),
                                  dT.stride(0), dT.stride(2), dT.stride(1), _stream())
    _lib.check(rc, 'mgp_agg_bwd_x')
    return dT


class _AggFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, T, G):
        ctx.save_for_backward(G)
        return agg_fwd(T, G)

    @staticmethod
    def backward(ctx, dY):
        (G,) = ctx.saved_tensors
        if ctx.needs_input_grad[1]:
            raise MgpError("gradient w.r.t. delay_gso is not supported (the reference never asks for it)")
        dT = agg_bwd_x(dY.contiguous(), G) if ctx.needs_input_grad[0] else None
        return dT, None


def aggregate(T, G):
    """Differentiable (w.r.t. T) graph-shift aggregation."""
    return _AggFn.apply(T, G)


# ------------------------------------------------------------------------------------ dense layer
def dense_fwd(inp, W2, bias, act):
    """inp (B,Cin,T,N) view, W2 (Cout,Cin), bias (Cout) -> (B,Cout,T,N) contiguous."""
    _dev(inp, 'input'); _dev(W2, 'weight'); _dev(bias, 'bias')
    B, Cin, T, N = inp.shape
    Cout = W2.shape[0]
    assert W2.shape == (Cout, Cin) and bias.shape == (Cout,)
    if inp.stride(3) != 1:
        inp = inp.contiguous()
    W2 = W2.contiguous()
    out = torch.empty((B, Cout, T, N), device=inp.device, dtype=torch.float32)
    rc = _lib.lib().mgp_dense_fwd(_ptr(inp), _ptr(W2), _ptr(bias.contiguous()), _ptr(out), B, Cin, Cout, T, N,
                                  inp.stride(0), inp.stride(1), inp.stride(2), act, _stream())
    _lib.check(rc, 'mgp_dense_fwd')
    return out


def dense_bwd(dOut, out, inp, W2, act, need_dinp):
    B, Cin, T, N = inp.shape
    Cout = W2.shape[0]
    if inp.stride(3) != 1:
        inp = inp.contiguous()
    W2 = W2.contiguous()
    dOut = dOut.contiguous()
    L = _lib.lib()
    ws = torch.empty((max(1, L.mgp_dense_bwd_workspace(B, Cin, Cout, T, N)),), device=inp.device,
                     dtype=torch.float32)
    dW = torch.empty((Cout, Cin), device=inp.device, dtype=torch.float32)
    db = torch.empty((Cout,), device=inp.device, dtype=torch.float32)
    dIn = torch.empty((B, Cin, T, N), device=inp.device, dtype=torch.float32) if need_dinp else None
    rc = L.mgp_dense_bwd(_ptr(dOut), _ptr(out), _ptr(inp), _ptr(W2), _ptr(dW), _ptr(db), _ptr(dIn),
                         B, Cin, Cout, T, N, inp.stride(0), inp.stride(1), inp.stride(2), act, _ptr(ws), _stream())
    _lib.check(rc, 'mgp_dense_bwd')
    return dW, db, dIn


class _DenseFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, inp, W2, bias, act):
        out = dense_fwd(inp, W2, bias, act)
        ctx.act = act
        ctx.save_for_backward(inp, W2, out)
        return out

    @staticmethod
    def backward(ctx, dOut):
        inp, W2, out = ctx.saved_tensors
        dW, db, dIn = dense_bwd(dOut, out, inp, W2, ctx.act, ctx.needs_input_grad[0])
        return dIn, dW, db, None


def dense(inp, W2, bias, act):
    """Differentiable per-agent dense layer (the reference's (step,1) Conv2d + optional tanh)."""
    return _DenseFn.apply(inp, W2, bias, act)


# ------------------------------------------------------------------------------------ state update
def gso_update(A, G_prev, X_t, Xd_prev, K):
    """A (B,N,N), G_prev (B,K,N,N)|None, X_t (B,F,N), Xd_prev (B,K,F,N)|None -> (G_next, Xd_next).
    Reference state_with_delay.py:44-53."""
    _dev(A, 'A'); _dev(X_t, 'X_t')
    B, N, _ = A.shape
    F = X_t.shape[1]
    has_prev = G_prev is not None
    if has_prev:
        _dev(G_prev, 'G_prev'); _dev(Xd_prev, 'Xd_prev')
        assert G_prev.shape == (B, K, N, N) and Xd_prev.shape == (B, K, F, N)
        G_prev = G_prev.contiguous(); Xd_prev = Xd_prev.contiguous()
    A = A.contiguous(); X_t = X_t.contiguous()
    G_next = torch.empty((B, K, N, N), device=A.device, dtype=torch.float32)
    Xd_next = torch.empty((B, K, F, N), device=A.device, dtype=torch.float32)
    rc = _lib.lib().mgp_gso_update(_ptr(A), _ptr(G_prev), _ptr(G_next), _ptr(X_t), _ptr(Xd_prev), _ptr(Xd_next),
                                   B, K, F, N, 1 if has_prev else 0, _stream())
    _lib.check(rc, 'mgp_gso_update')
    return G_next, Xd_next


def gso_update_into(A, G_prev, G_next, X_t, Xd_prev, Xd_next, has_prev=True):
    """In-place variant on preallocated ping-pong buffers (no allocation: HIP-graph capturable)."""
    B, K, N, _ = G_next.shape
    F = X_t.shape[1]
    if not (A.is_contiguous() and X_t.is_contiguous() and G_next.is_contiguous() and Xd_next.is_contiguous()):
        raise MgpError("mgp_gso_update reads A (B,N,N) and X_t (B,F,N) with dense batch strides: pass contiguous tensors "
                       "(got strides A %s, X_t %s)" % (tuple(A.stride()), tuple(X_t.stride())))
    rc = _lib.lib().mgp_gso_update(_ptr(A), _ptr(G_prev), _ptr(G_next), _ptr(X_t), _ptr(Xd_prev), _ptr(Xd_next),
                                   B, K, F, N, 1 if has_prev else 0, _stream())
    _lib.check(rc, 'mgp_gso_update')


def gso_advance(G_prev, G_next, Xd_prev, Xd_next, has_prev=True):
    """In-place state transition when the simulator already wrote A_t into G_next[:,1] and X_t into Xd_next[:,0]."""
    B, K, N, _ = G_next.shape
    F = Xd_next.shape[2]
    rc = _lib.lib().mgp_gso_advance(_ptr(G_prev), _ptr(G_next), _ptr(Xd_prev), _ptr(Xd_next), B, K, F, N,
                                    1 if has_prev else 0, _stream())
    _lib.check(rc, 'mgp_gso_advance')


def gso_powers(A, K):
    """A (B,N,N) -> (B,K,N,N): I, A, A@A, ...   (reference state_with_delay.py:38-41)."""
    _dev(A, 'A')
    B, N, _ = A.shape
    A = A.contiguous()
    P = torch.empty((B, K, N, N), device=A.device, dtype=torch.float32)
    rc = _lib.lib().mgp_gso_powers(_ptr(A), _ptr(P), B, K, N, _stream())
    _lib.check(rc, 'mgp_gso_powers')
    return P


# ------------------------------------------------------------------------------------ flocking sim
def flock_step(x, u, params, A=None, A64=None, feat=None, feat64=None, reward=None, expert=None, x_out=None):
    """In-place sim step on x (B,N,4) fp64; any output may be None.  See include/mgp.h.
    u: (B,N,2) contiguous, or the Actor's output (B,1,2,N) contiguous (consumed without a transpose)."""
    _dev(x, 'x', torch.float64)
    B, N, _ = x.shape
    assert x.is_contiguous()
    su_agent, su_axis = 2, 1
    if u is not None:
        _dev(u, 'u')
        assert u.is_contiguous()
        if u.shape == (B, 1, 2, N) or u.shape == (B, 2, N):
            su_agent, su_axis = 1, N
        else:
            assert u.shape == (B, N, 2), "u must be (B,N,2) or (B,1,2,N)"
    # A / feat may be strided batch views (e.g. delay_gso_next[:, 1], delay_state_next[:, 0])
    sAb = sFb = 0
    if A is not None:
        assert A.shape == (B, N, N) and A.stride(2) == 1 and A.stride(1) == N
        sAb = A.stride(0)
    if feat is not None:
        assert feat.shape == (B, 6, N) and feat.stride(2) == 1 and feat.stride(1) == N
        sFb = feat.stride(0)
    if x_out is not None:
        _dev(x_out, 'x_out', torch.float64)
        assert x_out.shape == x.shape and x_out.is_contiguous()
    rc = _lib.lib().mgp_flock_step(_ptr(x), _ptr(x_out), _ptr(u), su_agent, su_axis, _ptr(A), _ptr(A64), _ptr(feat), _ptr(feat64),
                                   _ptr(reward), _ptr(expert), sAb, sFb, ctypes.byref(params), B, N, _stream())
    _lib.check(rc, 'mgp_flock_step')


def flock_step_advance(x, x_out, u, params, G_prev, G_next, Xd_prev, Xd_next, has_prev, reward=None, expert=None):
    """Fused simulator step + delayed-GSO/delay-line transition (one launch).  Returns False when the shape is not
    covered by the fused kernel (the caller then uses flock_step + gso_advance)."""
    B, N, _ = x.shape
    K = G_next.shape[1]
    su_agent, su_axis = (1, N) if (u.shape == (B, 1, 2, N) or u.shape == (B, 2, N)) else (2, 1)
    rc = _lib.lib().mgp_flock_step_advance(_ptr(x), _ptr(x_out), _ptr(u), su_agent, su_axis, _ptr(G_prev), _ptr(G_next),
                                           _ptr(Xd_prev), _ptr(Xd_next), _ptr(reward), _ptr(expert),
                                           ctypes.byref(params), B, K, N, 1 if has_prev else 0, _stream())
    if rc == -5:
        return False
    _lib.check(rc, 'mgp_flock_step_advance')
    return True


def rollout_supported(dims, K, N):
    cd = (ctypes.c_int * len(dims))(*dims)
    return bool(_lib.lib().mgp_rollout_supported(cd, len(dims) - 1, K, N))


RO_ENTER_CARRY, RO_EXIT_CARRY, RO_SKIP_DENSE = 1, 2, 4          # include/mgp.h: MGP_RO_*


def rollout_steps(x, G, Xd, weights, biases, dims, params, T, action=None, rewards=None, image=None, carry=None, flags=0,
                  f32ref=False):
    """T closed-loop policy steps for every episode in ONE launch, state updated in place (mgp_rollout_steps_ex).
    f32ref: the checker build with the hidden layers on fp32 MFMA (mgp_rollout_f32ref_steps_ex; tests only).
    x (B,N,4) f64 | G (B,K,N,N) | Xd (B,K,6,N) | weights[l] (out, in*step) / biases[l] fp32 | rewards (B,T) f64.
    image: prebuilt weight image (rollout_image), or None (built from weights / biases inside the launch);
    carry (B, rollout_carry_bytes) uint8 + flags (RO_*): factored hand-over of the operator history, see include/mgp.h.
    Returns False when the shape is outside the resident kernel's coverage (nothing was launched)."""
    _dev(x, 'x', torch.float64); _dev(G, 'G'); _dev(Xd, 'Xd')
    B, N, _ = x.shape
    K = G.shape[1]
    assert G.shape == (B, K, N, N) and Xd.shape == (B, K, 6, N) and G.is_contiguous() and Xd.is_contiguous()
    assert x.is_contiguous()
    if rewards is not None:
        assert rewards.shape == (B, T) and rewards.dtype == torch.float64 and rewards.is_contiguous()
    if action is not None:
        assert action.shape == (B, 1, 2, N) and action.is_contiguous()
    if carry is not None:
        assert carry.dtype == torch.uint8 and carry.is_contiguous() and carry.shape == (B, rollout_carry_bytes(K, N))
    cd = (ctypes.c_int * len(dims))(*dims)
    wa = ba = None
    if image is None:
        Ws = [w.contiguous() for w in weights]
        bs = [b_.contiguous() for b_ in biases]
        wa = (ctypes.c_void_p * len(Ws))(*[w.data_ptr() for w in Ws])
        ba = (ctypes.c_void_p * len(bs))(*[b_.data_ptr() for b_ in bs])
    else:
        _dev(image, 'image')
    entry = _lib.lib().mgp_rollout_f32ref_steps_ex if f32ref else _lib.lib().mgp_rollout_steps_ex
    rc = entry(_ptr(x), _ptr(G), _ptr(Xd), wa, ba, cd, len(dims) - 1, _ptr(action), _ptr(rewards),
               ctypes.byref(params), B, K, N, int(T), _ptr(image), _ptr(carry), int(flags), _stream())
    if rc == -5 and image is None and not f32ref and rollout_supported(tuple(dims), K, N):
        # a build that streams part of its weight image from HBM every step (two 128-wide hidden layers: rollout_w128x2.hip) cannot
        # build the image inside the launch: build it here, once per call, and launch again
        image = rollout_image(weights, biases, tuple(dims), K, N)
        if image is not None:
            rc = entry(_ptr(x), _ptr(G), _ptr(Xd), None, None, cd, len(dims) - 1, _ptr(action), _ptr(rewards),
                       ctypes.byref(params), B, K, N, int(T), _ptr(image), _ptr(carry), int(flags), _stream())
    if rc == -5:
        return False
    _lib.check(rc, 'mgp_rollout_steps_ex')
    return True


def rollout_collect(x, G, Xd, dims, params, T, frames, expert_io, beta, episode, seed, age0, ring_step0, carry, flags,
                    weights=None, biases=None, image=None, rewards=None):
    """T DAGGER data-collection steps in ONE launch (mgp_rollout_collect): every step files its starting state as a compact
    frame into `frames` (a FrameReplay-like object with .feat/.bits/.label/.age rings laid out [ring_steps][B]) and is driven
    by the expert with probability beta[b] (counter-based coin), else by the policy.  Returns False if the shape is not
    covered (outside mgp_rollout_supported)."""
    _dev(x, 'x', torch.float64); _dev(G, 'G'); _dev(Xd, 'Xd'); _dev(expert_io, 'expert_io'); _dev(beta, 'beta')
    B, N, _ = x.shape
    K = G.shape[1]
    assert G.is_contiguous() and Xd.is_contiguous() and x.is_contiguous() and expert_io.is_contiguous()
    assert expert_io.shape == (B, 2, N) and beta.shape == (B,) and episode.shape == (B,) and episode.dtype == torch.int32
    S = frames.ring_steps
    assert frames.feat.shape == (S, B, 6, N) and frames.bits.shape == (S, B, N, 2 if N <= 128 else 4) and frames.label.shape == (S, B, 2, N)
    assert frames.age.shape == (S, B) and frames.age.dtype == torch.int32 and frames.bits.dtype == torch.int64
    if rewards is not None:
        assert rewards.shape == (B, T) and rewards.dtype == torch.float64 and rewards.is_contiguous()
    cl = _lib.MgpCollect(frames.feat.data_ptr(), frames.bits.data_ptr(), frames.label.data_ptr(), frames.age.data_ptr(),
                         expert_io.data_ptr(), beta.data_ptr(), episode.data_ptr(), int(seed) & 0xFFFFFFFF, int(age0),
                         int(ring_step0), int(S))
    cd = (ctypes.c_int * len(dims))(*dims)
    wa = ba = None
    if image is None:
        Ws = [w.contiguous() for w in weights]
        bs = [b_.contiguous() for b_ in biases]
        wa = (ctypes.c_void_p * len(Ws))(*[w.data_ptr() for w in Ws])
        ba = (ctypes.c_void_p * len(bs))(*[b_.data_ptr() for b_ in bs])
    rc = _lib.lib().mgp_rollout_collect(_ptr(x), _ptr(G), _ptr(Xd), wa, ba, cd, len(dims) - 1, _ptr(rewards),
                                        ctypes.byref(params), B, K, N, int(T), _ptr(image), _ptr(carry), int(flags),
                                        ctypes.byref(cl), _stream())
    if rc == -5:
        return False
    _lib.check(rc, 'mgp_rollout_collect')
    return True


def replay_gather(frames, idx, X, G, Y, mean_pooling, cursor=None, nb=1):
    """Minibatch (X (Bt,K,6,N), G (Bt,K,N,N), Y (Bt,1,2,N)) rebuilt from the frame ring for the frame indices
    idx[(cursor or 0) * Bt + i] (mgp_replay_gather); idx int64 on the device, cursor (1,) int32 on the device or None.
    nb > 1: `nb` consecutive minibatches in one launch, X / G / Y holding nb * Bt samples (mgp_replay_gather_many)."""
    _dev(X, 'X'); _dev(G, 'G'); _dev(Y, 'Y'); _dev(idx, 'idx', torch.int64)
    Bn, K, _, N = X.shape
    assert Bn % nb == 0
    Bt = Bn // nb
    assert X.is_contiguous() and G.is_contiguous() and Y.is_contiguous() and G.shape == (Bn, K, N, N) and Y.numel() == Bn * 2 * N
    S, lanes = frames.feat.shape[0], frames.feat.shape[1]
    if getattr(frames, 'wrow', None) is not None:               # frames of the factored path (N > 256): weights stored, NW words per row
        assert frames.bits.shape[-1] == _lib.lib().mgp_sparse_words(N)
        rc = _lib.lib().mgp_replay_gather_rows(_ptr(frames.feat), _ptr(frames.bits), _ptr(frames.wrow), _ptr(frames.label),
                                               _ptr(frames.age), _ptr(idx), _ptr(cursor), Bt, nb, lanes, S, K, N, _ptr(X),
                                               _ptr(G), _ptr(Y), _stream())
        _lib.check(rc, 'mgp_replay_gather_rows')
        return
    rc = _lib.lib().mgp_replay_gather_many(_ptr(frames.feat), _ptr(frames.bits), _ptr(frames.label), _ptr(frames.age), _ptr(idx),
                                           _ptr(cursor), Bt, nb, lanes, S, K, N, 1 if mean_pooling else 0, _ptr(X), _ptr(G),
                                           _ptr(Y), _stream())
    _lib.check(rc, 'mgp_replay_gather_many')


def replay_aggregate(frames, idx, Z, Y, mean_pooling, cursor=None, nb=1):
    """The aggregated first-layer input of `nb` minibatches straight from the frame ring (mgp_replay_aggregate):
    Z[s, f K + k] = x_{t-k} . A_t A_{t-1} .. A_{t-k+1} (reference actor.py:64-75 on the frame history, the operator slices never
    formed), Y = the labels.  Z (nb Bt, 6 K, N), Y (nb Bt, 1, 2, N); idx / cursor as in replay_gather."""
    _dev(Z, 'Z'); _dev(Y, 'Y'); _dev(idx, 'idx', torch.int64)
    Bn, FK, N = Z.shape
    K = frames.K
    assert FK == 6 * K and Bn % nb == 0 and Z.is_contiguous() and Y.is_contiguous() and Y.numel() == Bn * 2 * N
    S, lanes = frames.feat.shape[0], frames.feat.shape[1]
    wrow = getattr(frames, 'wrow', None)
    if wrow is not None:
        assert frames.bits.shape[-1] == _lib.lib().mgp_sparse_words(N)
    rc = _lib.lib().mgp_replay_aggregate(_ptr(frames.feat), _ptr(frames.bits), _ptr(wrow), _ptr(frames.label), _ptr(frames.age),
                                         _ptr(idx), _ptr(cursor), Bn // nb, nb, lanes, S, K, N, 1 if mean_pooling else 0,
                                         _ptr(Z), _ptr(Y), _stream())
    _lib.check(rc, 'mgp_replay_aggregate')


def rollout_image(weights, biases, dims, K, N):
    """Weight image of the episode-resident kernel for this policy (MFMA fragment order), built once by a tiny kernel:
    pass it to rollout_steps(image=...) for as long as the weights do not change.  None when the shape is not covered."""
    cd = (ctypes.c_int * len(dims))(*dims)
    L = _lib.lib()
    n = L.mgp_rollout_image_floats(cd, len(dims) - 1, K, N)
    if n <= 0:
        return None
    Ws = [w.contiguous() for w in weights]
    bs = [b_.contiguous() for b_ in biases]
    wa = (ctypes.c_void_p * len(Ws))(*[w.data_ptr() for w in Ws])
    ba = (ctypes.c_void_p * len(bs))(*[b_.data_ptr() for b_ in bs])
    image = torch.empty((n,), device=Ws[0].device, dtype=torch.float32)
    _lib.check(L.mgp_rollout_image(wa, ba, cd, len(dims) - 1, K, N, _ptr(image), _stream()), 'mgp_rollout_image')
    return image


def rollout_carry_bytes(K, N):
    return int(_lib.lib().mgp_rollout_carry_bytes(int(K), int(N)))


def rollout_carry_to_dense(carry, G, K):
    """delay_gso slices 1..K-1 of every episode from the factored history `carry` (in place in G (B,K,N,N))."""
    _dev(G, 'G')
    B, K_, N, _ = G.shape
    assert K_ == K and G.is_contiguous() and carry.dtype == torch.uint8 and carry.is_contiguous()
    _lib.check(_lib.lib().mgp_rollout_carry_to_dense(_ptr(carry), _ptr(G), B, K, N, _stream()), 'mgp_rollout_carry_to_dense')


def flock_controller(x, params, centralized=False, u=None, u64=None):
    _dev(x, 'x', torch.float64)
    B, N, _ = x.shape
    rc = _lib.lib().mgp_flock_controller(_ptr(x), _ptr(u), _ptr(u64), ctypes.byref(params),
                                         1 if centralized else 0, B, N, _stream())
    _lib.check(rc, 'mgp_flock_controller')


# ------------------------------------------------------------------------------------ DAGGER update
def mse_grad(pred, target, want_grad=True):
    """Returns (loss (1,) tensor, dPred or None):  F.mse_loss(pred, target) and its gradient."""
    _dev(pred, 'pred'); _dev(target, 'target')
    assert pred.shape == target.shape
    pred = pred.contiguous(); target = target.contiguous()
    loss = torch.empty((1,), device=pred.device, dtype=torch.float32)
    dPred = torch.empty_like(pred) if want_grad else None
    rc = _lib.lib().mgp_mse_grad(_ptr(pred), _ptr(target), _ptr(dPred), _ptr(loss), pred.numel(), _stream())
    _lib.check(rc, 'mgp_mse_grad')
    return loss, dPred


class _MseFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred, target):
        loss, dPred = mse_grad(pred, target, True)
        ctx.save_for_backward(dPred)
        return loss.reshape(())

    @staticmethod
    def backward(ctx, g):
        (dPred,) = ctx.saved_tensors
        return dPred * g, None


def mse_loss(pred, target):
    """Differentiable mean-squared error (reference gnn_dagger.py:91)."""
    return _MseFn.apply(pred, target)


def adam_step(param, grad, m, v, lr, step, beta1=0.9, beta2=0.999, eps=1e-8):
    """torch.optim.Adam defaults on one flat fp32 buffer, in place (reference gnn_dagger.py:49,93)."""
    for t, n in ((param, 'param'), (grad, 'grad'), (m, 'm'), (v, 'v')):
        _dev(t, n)
        assert t.is_contiguous()
    rc = _lib.lib().mgp_adam_step(_ptr(param), _ptr(grad), _ptr(m), _ptr(v), param.numel(),
                                  lr, beta1, beta2, eps, step, _stream())
    _lib.check(rc, 'mgp_adam_step')


def adam_step_dev(param, grad, m, v, lr, step_dev, beta1=0.9, beta2=0.999, eps=1e-8):
    """Same update with the step count on the device: `step_dev` (1,) int32 holds the number of steps taken so far; the
    kernel uses step_dev + 1 for the bias corrections and stores it back (graph-replayable, no host state)."""
    for t, n in ((param, 'param'), (grad, 'grad'), (m, 'm'), (v, 'v')):
        _dev(t, n)
        assert t.is_contiguous()
    _dev(step_dev, 'step_dev', torch.int32)
    rc = _lib.lib().mgp_adam_step_dev(_ptr(param), _ptr(grad), _ptr(m), _ptr(v), param.numel(),
                                      lr, beta1, beta2, eps, _ptr(step_dev), _stream())
    _lib.check(rc, 'mgp_adam_step_dev')


__all__ = ['MgpFlockParams', 'MgpError', 'aggregate', 'dense', 'agg_fwd', 'agg_bwd_x', 'dense_fwd', 'dense_bwd',
           'gso_update', 'gso_update_into', 'gso_powers', 'flock_step', 'flock_controller', 'mse_grad', 'mse_loss',
           'gso_advance', 'flock_step_advance', 'rollout_supported', 'rollout_steps', 'rollout_collect', 'replay_gather', 'rollout_image', 'rollout_carry_bytes', 'rollout_carry_to_dense', 'adam_step', 'adam_step_dev', 'ACT_NONE', 'ACT_TANH']
