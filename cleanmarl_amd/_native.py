"""ctypes binding of libcleanmarl_hip.so (the C-ABI declared in include/cleanmarl_hip.h).

The product path has NO fallback: if the shared library is missing or a call fails this raises.
PyTorch is used only for device memory and streams; every pointer handed to the library is a raw
``tensor.data_ptr()``.
"""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("CM_LIB_PATH") or os.path.join(HERE, "libcleanmarl_hip.so")  # CM_LIB_PATH: probe builds (tools/probes)

NUM_STATS = 8
STAT_PG, STAT_ENT, STAT_KL, STAT_CLIP, STAT_VLOSS, STAT_COUNT = range(6)
OPT_ADAM, OPT_ADAMW, OPT_SGD, OPT_RMSPROP = 0, 1, 2, 3

_p = C.c_void_p
_i = C.c_int
_l = C.c_int64
_d = C.c_double
_f = C.c_float
_u64 = C.c_uint64
_sz = C.c_size_t



class OptStep(C.Structure):
    """cm_opt_step_t of include/cleanmarl_hip.h: one optimiser step, fused behind a training pass by the *_train_step entry points."""
    _fields_ = [("params", _p), ("exp_avg", _p), ("exp_avg_sq", _p), ("out_norm", _p), ("scratch", _p),
                ("lr", _d), ("beta1", _d), ("beta2", _d), ("eps", _d), ("weight_decay", _d), ("max_norm", _d), ("grad_scale", _d),
                ("step", C.c_int32), ("opt_kind", C.c_int32), ("stats_out", _p)]


_po = C.POINTER(OptStep)

# name -> (restype, argtypes); mirrors include/cleanmarl_hip.h one to one
SIGNATURES = {
    "cm_last_error": (C.c_char_p, []),
    "cm_version": (_i, []),
    "cm_mfma_mode": (_i, []),
    "cm_set_option": (_i, [C.c_char_p, C.c_char_p]),
    "cm_get_option": (C.c_char_p, [C.c_char_p]),
    "cm_mlp_forward_ld": (_i, [_p, _l, _l, _i, _i, _i, _i, _p, _p, _p, _p, _sz, _p]),
    "cm_mlp_forward_solo_ld": (_i, [_p, _l, _l, _i, _i, _i, _i, _p, _p, _p, _p, _sz, _p]),
    "cm_w0_image_bytes": (_sz, [_i, _i]),
    "cm_policy_act_episode_ld": (_i, [_p, _l, _p, _l, _i, _i, _i, _i, _i, _p, _u64, _l, _p, _p, _p, _sz, _p]),
    "cm_ppo_actor_fwd_bwd_ld": (_i, [_p, _l, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _p, _d, _d, _p, _p, _sz, _p]),
    "cm_critic_fwd_bwd_ld": (_i, [_p, _l, _p, _p, _i, _i, _i, _i, _i, _i, _i, _p, _p, _p, _sz, _p]),
    "cm_critic_fwd_bwd_h0_ld": (_i, [_p, _l, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _p, _p, _p, _sz, _p]),
    "cm_value_pass_keep_h0_ld": (_i, [_p, _l, _l, _i, _i, _i, _p, _p, _p, _p, _sz, _p]),
    "cm_rollout_spread_ld": (_i, [_p, _i, _i, _i, _i, _u64, _u64, _l, _l, _p, _i, _i, _d, _p, _l, _p, _l, _p, _p, _p, _p]),
    "cm_shape_env_fill_ld": (_i, [_i, _i, _i, _i, _i, _i, _i, _d, _u64, _l, _l, _p, _l, _p, _l, _p, _p]),
    "cm_stream_create_low_priority": (_p, []),
    "cm_stream_destroy": (_i, [_p]),
    "cm_mlp_param_count": (_l, [_i, _i, _i, _i]),
    "cm_gru_param_count": (_l, [_i, _i, _i]),
    "cm_mlp_forward": (_i, [_p, _l, _i, _i, _i, _i, _p, _p, _p, _p]),
    "cm_mlp_forward_workspace_bytes": (_sz, [_l, _i, _i, _i, _i]),
    "cm_mlp_forward_ws": (_i, [_p, _l, _i, _i, _i, _i, _p, _p, _p, _p, _sz, _p]),
    "cm_policy_act": (_i, [_p, _l, _p, _l, _l, _i, _i, _i, _i, _p, _u64, _l, _i, _p, _p, _l, _p]),
    "cm_policy_act_workspace_bytes": (_sz, [_l, _i, _i, _i, _i]),
    "cm_policy_act_ws": (_i, [_p, _l, _p, _l, _l, _i, _i, _i, _i, _p, _d, _u64, _l, _i, _p, _p, _l, _p, _sz, _p]),
    "cm_policy_act_eps": (_i, [_p, _l, _p, _l, _l, _i, _i, _i, _i, _p, _d, _u64, _l, _i, _p, _p, _l, _p]),
    "cm_coma_build_inputs": (_i, [_p, _p, _p, _i, _i, _i, _i, _i, _i, _p, _p]),
    "cm_gather_taken": (_i, [_p, _p, _l, _i, _p, _p]),
    "cm_nstep_returns": (_i, [_p, _p, _p, _i, _i, _i, _d, _i, _p, _p]),
    "cm_mlp_split_workspace_bytes": (_sz, [_l, _i, _i, _i, _i]),
    "cm_qcritic_fwd_bwd": (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _p, _p, _p, _sz, _p]),
    "cm_coma_advantage_workspace_bytes": (_sz, [_i, _i, _i]),
    "cm_coma_advantage": (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _p, _p, _p, _sz, _p]),
    "cm_coma_normalize_adv": (_i, [_p, _p, _i, _i, _i, _p]),
    "cm_coma_actor_fwd_bwd": (_i, [_p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _p, _d, _p, _p, _sz, _p]),
    "cm_coma_critic_workspace_bytes": (_sz, [_i, _i, _i, _i, _i, _i, _i, _i, _i]),
    "cm_coma_q_forward": (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _p, _p, _p, _sz, _p]),
    "cm_coma_critic_fwd_bwd": (_i, [_p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _p, _p, _p, _sz, _p]),
    "cm_coma_actor_fwd_bwd_ld": (_i, [_p, _l, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _p, _d, _p, _p, _sz, _p]),
    "cm_coma_q_forward_ld": (_i, [_p, _l, _p, _l, _p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _p, _p, _p, _sz, _p]),
    "cm_coma_critic_fwd_bwd_ld": (_i, [_p, _l, _p, _l, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _p, _p, _p, _sz, _p]),
    "cm_polyak_update": (_i, [_p, _p, _l, _d, _p]),
    "cm_policy_act_greedy": (_i, [_p, _l, _p, _l, _l, _i, _i, _i, _i, _p, _p, _p, _l, _p]),
    "cm_policy_act_episode": (_i, [_p, _p, _l, _i, _i, _i, _i, _i, _p, _u64, _l, _p, _p, _p]),
    "cm_td_lambda_scan": (_i, [_p, _p, _p, _i, _i, _i, _i, _d, _d, _p, _p, _p]),
    "cm_masked_moments_workspace_bytes": (_sz, [_i, _i, _i]),
    "cm_masked_moments": (_i, [_p, _p, _i, _i, _i, _p, _p, _sz, _p]),
    "cm_normalize": (_i, [_p, _p, _i, _i, _i, _p, _f, _i, _p]),
    "cm_mlp_train_workspace_bytes": (_sz, [_i, _i, _i, _i]),
    "cm_ppo_actor_workspace_bytes": (_sz, [_i, _i, _i, _i, _i, _i, _i]),
    "cm_ppo_actor_fwd_bwd": (_i, [_p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _p, _d, _d, _p, _p, _sz, _p]),
    "cm_critic_workspace_bytes": (_sz, [_i, _i, _i, _i, _i, _i, _i]),
    "cm_critic_fwd_bwd": (_i, [_p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _p, _p, _p, _sz, _p]),
    "cm_grad_norm_clip_adam": (_i, [_p, _p, _p, _p, _l, _i, _d, _d, _d, _d, _d, _i, _d, _d, _p, _p]),
    "cm_opt_step_scratch_bytes": (_sz, []),
    "cm_optimizer_step": (_i, [_p, _l, _po, _p]),
    "cm_ppo_actor_train_step_ld": (_i, [_p, _l, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _d, _d, _p, _p, _sz, _po, _p]),
    "cm_critic_train_step_ld": (_i, [_p, _l, _p, _p, _i, _i, _i, _i, _i, _i, _i, _p, _p, _sz, _po, _p]),
    "cm_critic_train_step_h0_ld": (_i, [_p, _l, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _p, _p, _sz, _po, _p]),
    "cm_gru_actor_chunk_train_step": (_i, [_p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _p, _p, _d, _d, _p, _p, _sz, _po, _p]),
    "cm_peer_handle_bytes": (_sz, []),
    "cm_peer_mailbox_bytes": (_sz, [_i, _l]),
    "cm_peer_mailbox_alloc": (_i, [_sz, C.POINTER(_p), _p]),
    "cm_peer_mailbox_open": (_i, [_p, C.POINTER(_p)]),
    "cm_peer_mailbox_close": (_i, [_p]),
    "cm_peer_mailbox_free": (_i, [_p]),
    "cm_peer_push": (_i, [_p, _l, _i, _i, C.POINTER(_p), C.c_uint32, _p]),
    "cm_optimizer_step_peer": (_i, [_p, _l, _p, _i, C.c_uint32, _po, _d, _p, _p]),
    "cm_gru_workspace_bytes": (_sz, [_i, _i, _i, _i, _i, _i]),
    "cm_gru_actor_chunk_fwd_bwd": (_i, [_p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _p, _p, _p, _d, _d,
                                        _p, _p, _sz, _p]),
    "cm_gru_policy_act": (_i, [_p, _l, _p, _l, _l, _i, _i, _i, _p, _p, _u64, _l, _i, _p, _p, _l, _p]),
    "cm_gru_policy_act_workspace_bytes": (_sz, [_l, _i, _i, _i]),
    "cm_gru_policy_act_ws": (_i, [_p, _l, _p, _l, _l, _i, _i, _i, _p, _p, _d, _u64, _l, _i, _p, _p, _l, _p, _sz, _p]),
    "cm_synth_env_reset": (_i, [_p, _i, _i, _i, _u64, _l, _l, _p, _p, _i, _p]),
    "cm_synth_env_step": (_i, [_p, _p, _i, _i, _i, _i, _i, _p, _p, _p, _p]),
    "cm_shape_env_fill": (_i, [_i, _i, _i, _i, _i, _i, _i, _d, _u64, _l, _l, _p, _p, _p, _p]),
    "cm_shape_env_reward": (_i, [_i, _i, _i, _i, _u64, _l, _l, _p, _p, _p]),
    "cm_rollout_spread_supported": (_i, [_i, _i, _i, _i]),
    "cm_rollout_spread": (_i, [_p, _i, _i, _i, _i, _u64, _u64, _l, _l, _p, _i, _i, _p, _p, _p, _p, _p, _p]),
    "cm_gru_rollout_spread_supported": (_i, [_i, _i, _i]),
    "cm_gru_rollout_spread": (_i, [_p, _i, _i, _i, _i, _u64, _u64, _l, _l, _p, _i, _p, _p, _p, _p, _p, _p]),
    "cm_gru_rollout_spread_ld": (_i, [_p, _i, _i, _i, _i, _u64, _u64, _l, _l, _p, _i, _p, _p, _l, _p, _p, _p, _p]),
    "cm_rollout_spread_eps": (_i, [_p, _i, _i, _i, _i, _u64, _u64, _l, _l, _p, _i, _i, _d, _p, _p, _p, _p, _p, _p]),
    "cm_ppo_actor_issued_flop_per_row": (_d, [_i, _i, _i, _i]),
    "cm_clock_probe": (_i, [_p]),
    "cm_philox4x32_host": (_i, [_p, _l, _p]),
    "cm_philox4x32_device": (_i, [_p, _l, _p, _p]),
}

_lib = None


class NativeError(RuntimeError):
    pass


def load():
    """Load the shared library (once).  Raises if it has not been built -- there is no CPU fallback."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise NativeError(
            f"{LIB_PATH} not found: build it first (python -c 'import __graft_entry__ as g; g.build()' "
            "or python cleanmarl_amd/build.py).  cleanmarl_amd has no CPU fallback by design.")
    # torch wheels bundle their own libamdhip64: import torch FIRST so that our library binds to the same HIP
    # runtime instance that owns torch's streams and allocations (two runtimes in one process = "no device").
    import torch  # noqa: F401
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError here = header/library mismatch
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    sync_env_options()
    return lib


# The C library never reads the environment (include/cleanmarl_hip.h: cm_set_option).  The CM_* variables below are a convenience of
# THIS package for A/B runs and tests: they are mapped onto cm_set_option when the library is loaded and again at the entry of every
# learner update / target computation / rollout (sync_env_options is a few dictionary look-ups; the library is called on a change only).
ENV_OPTIONS = {"CM_MLP_FORMS": "mlp_forms", "CM_CRITIC_SCHEDULE": "critic_schedule", "CM_GRU_TILE": "gru_tile",
               "CM_ROLLOUT_TILE": "rollout_tile", "CM_MFMA": "mfma", "CM_WIDE_SCHEDULE": "wide_schedule", "CM_DW0_BATCH": "dw0_batch", "CM_DW0_GRID": "dw0_grid", "CM_TRAIN_GRID": "train_grid", "CM_TILE_SPLIT": "tile_split"}
_DEFAULTS = {"mlp_forms": "auto", "critic_schedule": "auto", "gru_tile": "auto", "rollout_tile": "auto", "mfma": "fp32", "wide_schedule": "auto", "dw0_batch": "auto", "dw0_grid": "auto", "train_grid": "auto", "tile_split": "auto"}
_applied = {}
_env_seen = {}


def set_option(key, value):
    """cm_set_option(key, value); raises on an unknown key / value."""
    check(load().cm_set_option(key.encode(), str(value).encode()), f"cm_set_option({key}, {value})")
    _applied[key] = str(value)


def get_option(key):
    v = load().cm_get_option(key.encode())
    if v is None:
        raise NativeError(f"unknown option {key!r}")
    return v.decode()


def sync_env_options():
    """Apply CM_* variables whose value CHANGED since the last look (an explicit set_option stays until its variable changes)."""
    for env, key in ENV_OPTIONS.items():
        cur = os.environ.get(env)
        if _env_seen.get(env, None) == cur and env in _env_seen:
            continue
        if _lib is None:
            return
        want = cur or _DEFAULTS[key]
        rc = _lib.cm_set_option(key.encode(), want.encode())
        if rc != 0:
            raise NativeError(f"{env}={want!r}: {(_lib.cm_last_error() or b'?').decode()}")
        _env_seen[env] = cur
        _applied[key] = want


def check(rc, what):
    if rc != 0:
        msg = load().cm_last_error()
        raise NativeError(f"{what} failed (rc={rc}): {msg.decode() if msg else '?'}")


def ptr(t):
    """Raw device pointer of a torch tensor (None -> NULL)."""
    if t is None:
        return None
    return C.c_void_p(t.data_ptr())


def stream_ptr(stream=None):
    import torch
    s = stream if stream is not None else torch.cuda.current_stream()
    return C.c_void_p(s.cuda_stream)


_LOW_PRIORITY = {}


def low_priority_stream(device):
    """torch view (ExternalStream) of a lowest-priority HIP stream created by the library: ONE per device and process, shared by every
    learner.  HIP multiplexes its streams onto a handful of hardware queues; a process that kept creating streams (bench.py builds a
    learner per workload) ended up with a "second" stream on the default stream's queue -- the critic's epochs then ran serialised
    with the actor's (1024-env share of config 3: 3.5 ms instead of 2.7)."""
    import torch
    key = torch.device(device).index or 0
    if key not in _LOW_PRIORITY:
        h = load().cm_stream_create_low_priority()
        if not h:
            raise NativeError("cm_stream_create_low_priority failed: " + (load().cm_last_error() or b"?").decode())
        _LOW_PRIORITY[key] = torch.cuda.ExternalStream(h, device=device)
    return _LOW_PRIORITY[key]


_SIDE = {}


def side_stream(device):
    """A normal-priority second stream, one per device and process (same reason as low_priority_stream)."""
    import torch
    key = torch.device(device).index or 0
    if key not in _SIDE:
        _SIDE[key] = torch.cuda.Stream(device=device)
    return _SIDE[key]
