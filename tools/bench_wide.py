#!/usr/bin/env python3
"""Layered schedule (csrc/cm_mlp_wide.h) vs the fused kernels at config-3 shapes: MAPPO learner (targets + 3 epochs) with
critic widths 64 / 128 / 256, a wide actor, and a COMA iteration with the reference's default 128-wide critic.
usage: python tools/bench_wide.py [--envs E]"""
import argparse
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cleanmarl_amd import _native as _N  # noqa: E402
if os.environ.get("CM_LIB"):  # A/B builds of the same sources (e.g. -DCM_WT_K=64)
    _N.LIB_PATH = os.path.abspath(os.environ["CM_LIB"])
from cleanmarl_amd.coma_learner import COMAHParams, COMALearner, coma_critic_input_dim  # noqa: E402
from cleanmarl_amd.learner import HParams, NetSpec, PPOLearner, init_params_like_torch  # noqa: E402
from cleanmarl_amd.rollout import SyntheticSpreadRollout  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--envs", type=int, default=4096)
args = ap.parse_args()
E, A, T = args.envs, 8, 128
dev = torch.device("cuda:0")
roll = SyntheticSpreadRollout(E, A, T, seed=1, agent_ids=True, device=dev)
Do, Ds, K = roll.Do, roll.Ds, roll.K


def timeit(fn, n=5, w=2):
    for _ in range(w):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


torch.manual_seed(1)
for (ah, al, ch, cl) in [(64, 1, 64, 1), (64, 1, 128, 1), (64, 1, 256, 1), (64, 1, 64, 3), (128, 1, 128, 1)]:
    aspec, cspec = NetSpec(Do, ah, al, K), NetSpec(Ds, ch, cl, 1)
    L = PPOLearner("mappo", aspec, cspec, A, HParams(), dev, init_params_like_torch(aspec), init_params_like_torch(cspec))
    fused = ah <= 64 and al <= 2
    b = roll.collect(L.actor, aspec)
    ms_r = timeit(lambda: roll.collect(L.actor, aspec))
    ms = timeit(lambda: L.train_iteration(b))
    print(f"MAPPO {E}x{A}x{T} actor {ah}x{al + 1} critic {ch}x{cl + 1}: rollout {ms_r:7.2f} ms ({'fused' if fused else 'per-step layered'}), "
          f"targets + 3 epochs {ms:7.2f} ms -> {E * A * T / (ms + ms_r) / 1e3:6.1f} M agent-env-steps/s", flush=True)
    del L, b
    torch.cuda.empty_cache()

Dc = coma_critic_input_dim(Do, Ds, A, K)
for ch in (64, 128):
    aspec, cspec = NetSpec(Do, 64, 1, K), NetSpec(Dc, ch, 1, K)
    L = COMALearner(aspec, cspec, A, COMAHParams(), dev, init_params_like_torch(aspec), init_params_like_torch(cspec))
    b = roll.collect(L.actor, aspec, eps=0.3)
    ms = timeit(lambda: L.train_iteration(b))
    print(f"COMA  {E}x{A}x{T} critic {ch}x2 ({'factored, fused' if ch <= 64 else ('factored, layered' if os.environ.get('CM_WIDE_SCHEDULE') == 'layered' else 'factored, fused 128-wide tile')}): targets + critic + actor step "
          f"{ms:7.2f} ms", flush=True)
    del L, b
    torch.cuda.empty_cache()
