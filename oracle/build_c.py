"""Compile the plain-C part of the oracle (gcc) into oracle/_build/ (git-ignored; ships to the GPU box like any .so)."""
import ctypes
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "_build", "libtdlambda.so")


SRCS = ["td_lambda.c", "ppo_loss.c", "optim_moments.c"]


def build():
    srcs = [os.path.join(HERE, f) for f in SRCS]
    if os.path.exists(OUT) and all(os.path.getmtime(OUT) >= os.path.getmtime(f) for f in srcs):
        return OUT
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    subprocess.run(["gcc", "-O2", "-ffp-contract=off", "-shared", "-fPIC", "-o", OUT] + srcs + ["-lm"], check=True)
    return OUT


def td_lambda_c(reward, values, mask, gamma, lam):
    """numpy in / numpy out wrapper: reward [B,T] f32, values [B,T,A] f32, mask [B,T] bool."""
    import numpy as np
    lib = ctypes.CDLL(build())
    B, T, A = values.shape
    r = np.ascontiguousarray(reward, np.float32); v = np.ascontiguousarray(values, np.float32)
    m = np.ascontiguousarray(mask, np.uint8)
    ret = np.empty((B, T, A), np.float32); adv = np.empty((B, T, A), np.float32)
    P = ctypes.c_void_p
    lib.td_lambda_ref.argtypes = [P, P, P, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_double, ctypes.c_double, P, P]
    lib.td_lambda_ref(r.ctypes.data, v.ctypes.data, m.ctypes.data, B, T, A, gamma, lam, ret.ctypes.data, adv.ctypes.data)
    return ret, adv


def ppo_losses_c(batch, actor_params, critic_params, adv, ret, ppo_clip, entropy_coef, algo):
    """Epoch-1 logged scalars of the reference's loss loop from the plain-C restatement (oracle/ppo_loss.c).
    batch: reference-layout torch tensors; params: lists of torch tensors in parameters() order."""
    import numpy as np
    lib = ctypes.CDLL(build())
    f32 = lambda t: np.ascontiguousarray(t.detach().numpy() if hasattr(t, "detach") else t, np.float32)
    obs, states = f32(batch["obs"]), f32(batch["states"])
    avail = np.ascontiguousarray(batch["avail"].numpy(), np.uint8); mask = np.ascontiguousarray(batch["mask"].numpy(), np.uint8)
    actions = np.ascontiguousarray(batch["actions"].numpy(), np.int64); lp = f32(batch["log_probs"])
    adv, ret = f32(adv), f32(ret)
    flat = lambda ps: np.concatenate([f32(p).reshape(-1) for p in ps])
    ap, cp = flat(actor_params), flat(critic_params)
    B, T, A, Do = obs.shape
    Ds, K = states.shape[-1], avail.shape[-1]
    out = np.zeros(6, np.float64)
    P, I, D = ctypes.c_void_p, ctypes.c_int, ctypes.c_double
    lib.ppo_losses_ref.argtypes = [P] * 8 + [I] * 6 + [P, I, I, P, I, I, I, D, D, P]
    lib.ppo_losses_ref(obs.ctypes.data, states.ctypes.data, avail.ctypes.data, actions.ctypes.data, lp.ctypes.data, adv.ctypes.data,
                       ret.ctypes.data, mask.ctypes.data, B, T, A, Do, Ds, K, ap.ctypes.data, actor_params[0].shape[0],
                       len(actor_params) // 2 - 2, cp.ctypes.data, critic_params[0].shape[0], len(critic_params) // 2 - 2,
                       1 if algo == "ippo" else 0, float(ppo_clip), float(entropy_coef), out.ctypes.data)
    return dict(actor_loss=out[0], critic_loss=out[1], entropy=out[2], kl=out[3], clipfrac=out[4], n_valid=out[5])


def _lib():
    return ctypes.CDLL(build())


def clip_optim_step_c(params, grads, m, v, step, lr, kind, max_norm, beta1=0.9, beta2=0.999, eps=1e-8, weight_decay=0.0):
    """oracle/optim_moments.c::clip_optim_step_ref on flat float32 numpy arrays (updated IN PLACE); returns the pre-clip norm.
    kind: "Adam" | "AdamW" | "SGD" | "RMSprop" with torch's defaults (AdamW: weight_decay 0.01, RMSprop: alpha 0.99)."""
    import numpy as np
    kinds = {"Adam": 0, "AdamW": 1, "SGD": 2, "RMSprop": 3}
    if kind == "AdamW" and weight_decay == 0.0:
        weight_decay = 0.01
    if kind == "RMSprop":
        beta2 = 0.99
    lib = _lib()
    P, D, I = ctypes.c_void_p, ctypes.c_double, ctypes.c_int
    lib.clip_optim_step_ref.restype = D
    lib.clip_optim_step_ref.argtypes = [P, P, P, P, ctypes.c_long, I, D, D, D, D, D, I, D]
    for a in (params, grads, m, v):
        assert a.dtype == np.float32 and a.flags["C_CONTIGUOUS"]
    return lib.clip_optim_step_ref(params.ctypes.data, grads.ctypes.data, m.ctypes.data, v.ctypes.data, params.size, int(step), float(lr),
                                   beta1, beta2, eps, weight_decay, kinds[kind], float(max_norm))


def masked_normalize_c(x, mask, eps=0.0, valid_only=False):
    """oracle/optim_moments.c::masked_normalize_ref: x [B,T,A] float32 (copy returned), mask [B,T] -> (normalised x, (count, mean, std))."""
    import numpy as np
    x = np.ascontiguousarray(x, np.float32).copy(); mk = np.ascontiguousarray(mask, np.uint8)
    B, T, A = x.shape
    out = np.zeros(3, np.float64)
    lib = _lib()
    lib.masked_normalize_ref.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_double,
                                         ctypes.c_int, ctypes.c_void_p]
    lib.masked_normalize_ref(x.ctypes.data, mk.ctypes.data, B, T, A, float(eps), int(valid_only), out.ctypes.data)
    return x, tuple(out)


def gru_cell_c(x, h, W_ih, W_hh, b_ih, b_hh):
    """oracle/optim_moments.c::gru_cell_ref row by row: x [R,I], h [R,H] -> h' [R,H]."""
    import numpy as np
    f = lambda a: np.ascontiguousarray(a, np.float32)
    x, h, W_ih, W_hh, b_ih, b_hh = map(f, (x, h, W_ih, W_hh, b_ih, b_hh))
    R, I = x.shape; H = h.shape[1]
    out = np.empty((R, H), np.float32)
    lib = _lib()
    P = ctypes.c_void_p
    lib.gru_cell_ref.argtypes = [P] * 6 + [ctypes.c_int, ctypes.c_int, P]
    for r in range(R):
        lib.gru_cell_ref(x[r].ctypes.data, h[r].ctypes.data, W_ih.ctypes.data, W_hh.ctypes.data, b_ih.ctypes.data, b_hh.ctypes.data, I, H,
                         out[r].ctypes.data)
    return out


def categorical_sample_c(logits, u):
    """oracle/optim_moments.c::categorical_sample_ref row by row: masked logits [R,K], uniforms [R] -> (actions, logp)."""
    import numpy as np
    z = np.ascontiguousarray(logits, np.float32); u = np.ascontiguousarray(u, np.float32)
    R, K = z.shape
    act = np.empty(R, np.int64); lp = np.empty(R, np.float32)
    lib = _lib()
    lib.categorical_sample_ref.restype = ctypes.c_int
    lib.categorical_sample_ref.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_float, ctypes.c_void_p]
    one = ctypes.c_float()
    for r in range(R):
        act[r] = lib.categorical_sample_ref(z[r].ctypes.data, K, float(u[r]), ctypes.byref(one))
        lp[r] = one.value
    return act, lp
