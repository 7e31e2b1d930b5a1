"""Compile the plain-C part of the oracle (gcc) into oracle/_build/ (git-ignored; ships to the GPU box like any .so)."""
import ctypes
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "_build", "libtdlambda.so")


def build():
    src = os.path.join(HERE, "td_lambda.c")
    if os.path.exists(OUT) and os.path.getmtime(OUT) >= os.path.getmtime(src):
        return OUT
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    subprocess.run(["gcc", "-O2", "-ffp-contract=off", "-shared", "-fPIC", "-o", OUT, src], check=True)
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
