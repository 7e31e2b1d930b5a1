"""Compile the plain-C part of the oracle (gcc) into oracle/_build/ (git-ignored; ships to the GPU box like any .so)."""
import ctypes
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "_build", "libtdlambda.so")


SRCS = ["td_lambda.c", "ppo_loss.c"]


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
