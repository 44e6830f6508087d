"""ctypes binding of libmgp.so (the C ABI declared in include/mgp.h).

The product has NO CPU fallback: if the library cannot be loaded every op raises.  Loading the
library itself needs no GPU (the CPU test-suite checks that it loads and exports every symbol).
"""
import ctypes
import os
import re
import threading

from . import build as _build

_c_float_p = ctypes.POINTER(ctypes.c_float)
_c_double_p = ctypes.POINTER(ctypes.c_double)
_vp = ctypes.c_void_p
_int = ctypes.c_int
_long = ctypes.c_long
_f32 = ctypes.c_float


class MgpFlockParams(ctypes.Structure):
    """Mirror of struct MgpFlockParams (include/mgp.h)."""
    _fields_ = [('comm_radius2', ctypes.c_double), ('dt', ctypes.c_double), ('action_gain', ctypes.c_double),
                ('max_accel', ctypes.c_double), ('ctrl_gain', ctypes.c_double), ('ctrl_clip', ctypes.c_double),
                ('reward_scale', ctypes.c_double), ('mean_pooling', ctypes.c_int), ('n_leaders', ctypes.c_int),
                ('centralized', ctypes.c_int), ('link_drop', ctypes.c_uint), ('link_seed', ctypes.c_uint),
                ('reserved_', ctypes.c_int)]


class MgpCollect(ctypes.Structure):
    """Mirror of struct MgpCollect (include/mgp.h): frame ring + coin parameters of a collecting rollout."""
    _fields_ = [('feat', ctypes.c_void_p), ('bits', ctypes.c_void_p), ('label', ctypes.c_void_p), ('age', ctypes.c_void_p),
                ('expert_io', ctypes.c_void_p), ('beta', ctypes.c_void_p), ('episode', ctypes.c_void_p),
                ('seed', ctypes.c_uint), ('age0', ctypes.c_int), ('ring_step0', ctypes.c_int), ('ring_steps', ctypes.c_int)]


class MgpSparseCollect(ctypes.Structure):
    """Mirror of struct MgpSparseCollect (include/mgp.h): one collected step on the factored state (N > 256)."""
    _fields_ = [('feat', ctypes.c_void_p), ('bits', ctypes.c_void_p), ('wrow', ctypes.c_void_p), ('label', ctypes.c_void_p),
                ('age', ctypes.c_void_p), ('expert', ctypes.c_void_p), ('beta', ctypes.c_void_p), ('episode', ctypes.c_void_p),
                ('seed', ctypes.c_uint), ('age_now', ctypes.c_int), ('ring_step', ctypes.c_int), ('ring_steps', ctypes.c_int)]


# name -> (restype, argtypes).  Pointers are passed as raw integers (tensor.data_ptr()).
SIGNATURES = {
    'mgp_version': (_int, []),
    'mgp_strerror': (ctypes.c_char_p, [_int]),
    'mgp_last_hip_error': (ctypes.c_char_p, []),
    'mgp_device_info': (_int, [ctypes.c_char_p, _int]),
    'mgp_set_launch_events': (_int, [_vp, _vp]),
    'mgp_agg_fwd': (_int, [_vp, _vp, _vp, _int, _int, _int, _int, _long, _long, _long, _long, _long, _long, _vp]),
    'mgp_agg_bwd_x': (_int, [_vp, _vp, _vp, _int, _int, _int, _int, _long, _long, _long, _long, _long, _long, _vp]),
    'mgp_dense_fwd': (_int, [_vp, _vp, _vp, _vp, _int, _int, _int, _int, _int, _long, _long, _long, _int, _vp]),
    'mgp_dense_bwd_workspace': (_long, [_int, _int, _int, _int, _int]),
    'mgp_dense_bwd': (_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _int, _int, _int, _int, _int, _long, _long, _long,
                             _int, _vp, _vp]),
    'mgp_actor_saved_floats': (_long, [ctypes.POINTER(_int), _int, _int, _int, _int]),
    'mgp_actor_supported': (_int, [ctypes.POINTER(_int), _int, _int, _int]),
    'mgp_actor_fwd': (_int, [_vp, _vp, ctypes.POINTER(_vp), ctypes.POINTER(_vp), ctypes.POINTER(_int), _int,
                             _vp, _vp, _int, _int, _int, _vp]),
    'mgp_actor_deep_supported': (_int, [ctypes.POINTER(_int), _int, _int, _int]),
    'mgp_actor_fwd_deep': (_int, [_vp, _vp, ctypes.POINTER(_vp), ctypes.POINTER(_vp), ctypes.POINTER(_int), _int,
                                  _vp, _int, _int, _int, _vp]),
    'mgp_actor_bwd_workspace': (_long, [ctypes.POINTER(_int), _int, _int, _int, _int]),
    'mgp_actor_bwd': (_int, [_vp, _vp, ctypes.POINTER(_vp), ctypes.POINTER(_int), _int, ctypes.POINTER(_vp),
                             ctypes.POINTER(_vp), _int, _int, _int, _vp, _vp]),
    'mgp_gso_update': (_int, [_vp, _vp, _vp, _vp, _vp, _vp, _int, _int, _int, _int, _int, _vp]),
    'mgp_gso_advance': (_int, [_vp, _vp, _vp, _vp, _int, _int, _int, _int, _int, _vp]),
    'mgp_gso_powers': (_int, [_vp, _vp, _int, _int, _int, _vp]),
    'mgp_flock_step': (_int, [_vp, _vp, _vp, _long, _long, _vp, _vp, _vp, _vp, _vp, _vp, _long, _long,
                             ctypes.POINTER(MgpFlockParams), _int, _int, _vp]),
    'mgp_flock_step_advance': (_int, [_vp, _vp, _vp, _long, _long, _vp, _vp, _vp, _vp, _vp, _vp,
                                     ctypes.POINTER(MgpFlockParams), _int, _int, _int, _int, _vp]),
    'mgp_rollout_supported': (_int, [_vp, _int, _int, _int]),
    'mgp_rollout_steps': (_int, [_vp, _vp, _vp, _vp, _vp, _vp, _int, _vp, _vp, ctypes.POINTER(MgpFlockParams),
                                _int, _int, _int, _int, _vp]),
    'mgp_rollout_steps_ex': (_int, [_vp, _vp, _vp, _vp, _vp, _vp, _int, _vp, _vp, ctypes.POINTER(MgpFlockParams),
                                   _int, _int, _int, _int, _vp, _vp, _int, _vp]),
    'mgp_rollout_collect': (_int, [_vp, _vp, _vp, _vp, _vp, _vp, _int, _vp, ctypes.POINTER(MgpFlockParams),
                                  _int, _int, _int, _int, _vp, _vp, _int, _vp, _vp]),
    'mgp_replay_gather': (_int, [_vp, _vp, _vp, _vp, _vp, _vp, _int, _int, _int, _int, _int, _int, _vp, _vp, _vp, _vp]),
    'mgp_replay_gather_many': (_int, [_vp, _vp, _vp, _vp, _vp, _vp, _int, _int, _int, _int, _int, _int, _int, _vp, _vp, _vp, _vp]),
    'mgp_rollout_image_floats': (_long, [_vp, _int, _int, _int]),
    'mgp_rollout_image': (_int, [_vp, _vp, _vp, _int, _int, _int, _vp, _vp]),
    'mgp_rollout_f32ref_supported': (_int, [_vp, _int, _int, _int]),
    'mgp_rollout_f32ref_steps_ex': (_int, [_vp, _vp, _vp, _vp, _vp, _vp, _int, _vp, _vp, ctypes.POINTER(MgpFlockParams),
                                          _int, _int, _int, _int, _vp, _vp, _int, _vp]),
    'mgp_rollout_f32ref_image_floats': (_long, [_vp, _int, _int, _int]),
    'mgp_rollout_f32ref_image': (_int, [_vp, _vp, _vp, _int, _int, _int, _vp, _vp]),
    'mgp_rollout_carry_bytes': (_long, [_int, _int]),
    'mgp_rollout_carry_to_dense': (_int, [_vp, _vp, _int, _int, _int, _vp]),
    'mgp_flock_controller': (_int, [_vp, _vp, _vp, ctypes.POINTER(MgpFlockParams), _int, _int, _int, _vp]),
    'mgp_mse_grad': (_int, [_vp, _vp, _vp, _vp, _long, _vp]),
    'mgp_adam_step': (_int, [_vp, _vp, _vp, _vp, _long, _f32, _f32, _f32, _f32, _int, _vp]),
    'mgp_adam_step_dev': (_int, [_vp, _vp, _vp, _vp, _long, _f32, _f32, _f32, _f32, _vp, _vp]),
    'mgp_train_supported': (_int, [_vp, _int, _int, _int, _int]),
    'mgp_train_workspace': (_long, [_vp, _int, _int, _int, _int]),
    'mgp_train_grads': (_int, [_vp, _vp, _vp, _vp, _vp, _vp, _int, _vp, _vp, _vp, _int, _int, _int, _vp]),
    'mgp_train_step_indexed': (_int, [_vp, _vp, _vp, _vp, _vp, _vp, _int, _vp, _vp, _vp, _vp, _vp, _int,
                                      _f32, _f32, _f32, _f32, _vp, _vp, _int, _int, _int, _vp]),
    'mgp_p2p_create': (_int, [_int, _int, _int, ctypes.POINTER(_vp)]),
    'mgp_p2p_handle_bytes': (_int, []),
    'mgp_p2p_handle': (_int, [_vp, _vp]),
    'mgp_p2p_connect': (_int, [_vp, _vp]),
    'mgp_p2p_set_timeout_ms': (_int, [_vp, _int]),
    'mgp_p2p_info': (_int, [_vp, ctypes.POINTER(_int), ctypes.POINTER(_int), ctypes.POINTER(_int), ctypes.POINTER(_int)]),
    'mgp_p2p_status': (_int, [_vp, ctypes.POINTER(_int), ctypes.POINTER(_int), _vp]),
    'mgp_p2p_destroy': (_int, [_vp]),
    'mgp_p2p_allreduce_mean': (_int, [_vp, _vp, _int, _vp]),
    'mgp_train_step_p2p': (_int, [_vp, _vp, _vp, _vp, _vp, _vp, _int, _vp, _vp, _vp, _vp, _vp, _int,
                                  _f32, _f32, _f32, _f32, _vp, _vp, _vp, _int, _int, _int, _vp, _vp]),
    'mgp_adam_step_filed': (_int, [_vp, _vp, _vp, _vp, _long, _f32, _f32, _f32, _f32, _vp, _vp, _vp, _int, _vp, _vp]),
    'mgp_sparse_words': (_int, [_int]),
    'mgp_flock_step_sparse': (_int, [_vp, _vp, _vp, _long, _long, _vp, _long, _vp, _long, _vp, _long, _vp, _vp, _vp,
                                     _int, _int, _vp]),
    'mgp_flock_step_cells': (_int, [_vp, _vp, _vp, _long, _long, _vp, _long, _vp, _long, _vp, _long, _vp, _vp, _vp,
                                    _int, _int, _vp]),
    'mgp_sparse_policy_supported': (_int, [_vp, _int, _int, _int]),
    'mgp_sparse_policy_image_floats': (_long, [_vp, _int, _int]),
    'mgp_sparse_policy_image': (_int, [_vp, _vp, _vp, _int, _int, _vp, _vp]),
    'mgp_sparse_policy_step': (_int, [_vp, _vp, _vp, _vp, _vp, _int, _vp, _vp, _int, _int, _int, _int, _int, _vp]),
    'mgp_sparse_to_dense': (_int, [_vp, _vp, _vp, _int, _int, _int, _int, _vp]),
    'mgp_sparse_force_direct': (_int, [_int]),
    'mgp_sparse_rollout_persistent': (_int, [_vp, _int, _int, _int, _vp]),
    'mgp_sparse_rollout_status': (_int, [_vp, _int, _int, _int, _vp]),
    'mgp_sparse_rollout': (_int, [_vp, _vp, _vp, _vp, _vp, _int, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _int, _int, _int, _int,
                                  ctypes.POINTER(_int), ctypes.POINTER(_int), _vp, _vp, _vp]),
    'mgp_flock_step_cells_nbr': (_int, [_vp, _vp, _vp, _long, _long, _vp, _long, _vp, _long, _vp, _long, _vp, _long, _vp, _vp,
                                        _vp, _int, _int, _vp]),
    'mgp_sparse_policy_collect': (_int, [_vp, _vp, _vp, _vp, _vp, _int, _vp, _vp, _int, _int, _int, _int, _int, _vp, _vp]),
    'mgp_replay_gather_rows': (_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _int, _int, _int, _int, _int, _int, _vp, _vp, _vp, _vp]),
    'mgp_flock_reset_check': (_int, [_vp, _int, _int, ctypes.c_double, _vp, _vp, _vp]),
    'mgp_replay_aggregate': (_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _int, _int, _int, _int, _int, _int, _int, _vp, _vp, _vp]),
    'mgp_train_agg_supported': (_int, [_vp, _int, _int, _int, _int]),
    'mgp_train_grads_agg': (_int, [_vp, _vp, _vp, _vp, _vp, _int, _vp, _vp, _vp, _int, _int, _int, _vp]),
    'mgp_train_step_agg': (_int, [_vp, _vp, _vp, _vp, _vp, _int, _vp, _vp, _vp, _vp, _vp, _int,
                                  _f32, _f32, _f32, _f32, _vp, _vp, _vp, _int, _int, _int, _vp, _vp]),
    'mgp_train_step': (_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _int, _f32, _f32, _f32, _f32, _vp,
                              _vp, _vp, _int, _int, _int, _vp]),
}

_lock = threading.Lock()
_lib = None


class MgpError(RuntimeError):
    pass


def header_symbols():
    """Names of every function include/mgp.h declares (used by the CPU symbol test)."""
    hdr = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'include', 'mgp.h')
    with open(hdr) as f:
        text = f.read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(mgp_[a-z0-9_]+)\s*\(', text)))


def lib():
    """Load (once) and return the ctypes handle.  Builds the library if it is missing and hipcc exists."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is not None:
            return _lib
        # torch bundles its own HIP runtime (soname libamdhip64.so.7).  It must be mapped BEFORE libmgp.so so
        # that our DT_NEEDED libamdhip64.so.7 binds to the very same runtime instance torch uses; loading
        # libmgp first would pull /opt/rocm's copy and leave two runtimes in one process (kernels launched
        # into a runtime that never saw torch's device context: "no ROCm-capable device").
        import torch  # noqa: F401
        path = _build.LIB_PATH
        if not _build.is_current():
            # missing, or older than the sources / flags it was built from: rebuild (a no-op for the other ranks of a
            # torchrun launch, which wait on the build lock and then find the stamp current)
            if _build.hipcc_path() is not None:
                try:
                    _build.build(verbose=False)
                except Exception as e:  # loud, never a silent fallback
                    raise MgpError("libmgp.so is missing or stale and could not be built: %s" % e)
            elif not os.path.exists(path):
                raise MgpError("libmgp.so is missing and hipcc is not available to build it")
            else:
                import warnings
                warnings.warn("libmgp.so does not match the current sources (csrc/build/libmgp.srchash) and hipcc is "
                              "not available to rebuild it: running the stale library", RuntimeWarning)
        try:
            handle = ctypes.CDLL(path)
        except OSError as e:
            raise MgpError("cannot load %s: %s" % (path, e))
        for name, (res, args) in SIGNATURES.items():
            try:
                fn = getattr(handle, name)
            except AttributeError:
                raise MgpError("libmgp.so does not export %s (stale build? run python -m "
                               "multiagent_gnn_policies_amd.build --force)" % name)
            fn.restype = res
            fn.argtypes = args
        if handle.mgp_version() <= 0:
            raise MgpError("libmgp.so reports an invalid version")
        _lib = handle
        return _lib


def check(rc, what):
    if rc != 0:
        msg = lib().mgp_strerror(rc)
        text = msg.decode() if msg else '?'
        if rc == -3:
            hip = lib().mgp_last_hip_error()
            text += ' [HIP: %s]' % (hip.decode() if hip else '?')
        raise MgpError("%s failed: %s (code %d)" % (what, text, rc))


def strerror(rc):
    return lib().mgp_strerror(rc).decode()
