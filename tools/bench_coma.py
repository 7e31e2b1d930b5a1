#!/usr/bin/env python3
"""COMA iteration on the on-device synthetic env at config-3 shapes (4096 envs x 8 agents x 128 steps unless --envs):
eps-mixed rollout, targets from the target critic, one critic step, polyak, one actor step.  Prints ONE JSON line in the
style of bench.py (metric = agent-env-steps/s of the whole iteration) with per-phase and per-kernel-group times, the
roofline of the dominant kernel group (Q-critic fwd+bwd) and a bounded CPU baseline (oracle/coma.py, kind "port").
usage: python tools/bench_coma.py [--envs E] [--steps K] [--warmup W] [--no-cpu-baseline]"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cleanmarl_amd import _native as N  # noqa: E402
from cleanmarl_amd.coma_learner import COMAHParams, COMALearner, coma_critic_input_dim  # noqa: E402
from cleanmarl_amd.learner import NetSpec, init_params_like_torch  # noqa: E402
from cleanmarl_amd.rollout import SyntheticSpreadRollout  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--envs", type=int, default=4096)
ap.add_argument("--agents", type=int, default=8)
ap.add_argument("--T", type=int, default=128)
ap.add_argument("--steps", type=int, default=5)
ap.add_argument("--warmup", type=int, default=2)
ap.add_argument("--no-cpu-baseline", action="store_true")
ap.add_argument("--cpu-envs", type=int, default=16)
ap.add_argument("--critic-hidden", type=int, default=64, help="critic width (the reference's default is 128: coma_multienvs.py:35)")
args = ap.parse_args()
E, A, T = args.envs, args.agents, args.T
dev = torch.device("cuda:0")
roll = SyntheticSpreadRollout(E, A, T, seed=1, agent_ids=True, device=dev)  # padded rows: the COMA entry points take leading dimensions
Do, Ds, K = roll.Do, roll.Ds, roll.K
Dc = coma_critic_input_dim(Do, Ds, A, K)
Hc = args.critic_hidden
aspec, cspec = NetSpec(Do, 64, 1, K), NetSpec(Dc, Hc, 1, K)
torch.manual_seed(1)
L = COMALearner(aspec, cspec, A, COMAHParams(), dev, init_params_like_torch(aspec), init_params_like_torch(cspec))
ev = lambda: torch.cuda.Event(enable_timing=True)


def one(evts=None):
    e = [ev() for _ in range(4)]
    e[0].record()
    b = roll.collect(L.actor, aspec, eps=0.3)
    e[1].record()
    L.compute_targets(b)
    e[2].record()
    L.update(b)
    e[3].record()
    if evts is not None:
        evts.append(e)


for _ in range(args.warmup):
    one()
torch.cuda.synchronize()
evts = []
t0 = time.perf_counter()
for _ in range(args.steps):
    one(evts)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
ph = [sum(e[i].elapsed_time(e[i + 1]) for e in evts) / len(evts) for i in range(3)]

# per-kernel-group timing of the update (separate pass so the events do not perturb the iteration time)
b = roll.collect(L.actor, aspec, eps=0.3)
L.compute_targets(b)
lib, s = L.lib, N.stream_ptr()
rows = E * A * T


def timed(fn, n=3):
    fn(); torch.cuda.synchronize()
    a, z = ev(), ev()
    a.record()
    for _ in range(n):
        fn()
    z.record(); torch.cuda.synchronize()
    return a.elapsed_time(z) / n


k = {}
k["q_forward"] = timed(lambda: L._q(L.target, b.avail, L.q, b, s))
k["critic_fwd_bwd"] = timed(lambda: N.check(lib.cm_coma_critic_fwd_bwd(N.ptr(b.state), N.ptr(b.obs), N.ptr(b.action), N.ptr(b.ret), N.ptr(b.ep_len), E, A, T, Ds, Do, K,
                                                                        Hc, 1, N.ptr(L.critic), N.ptr(L.g_critic), N.ptr(L.ws), L.ws.numel(), s), "c"))
k["actor_forward"] = timed(lambda: N.check(lib.cm_mlp_forward(N.ptr(b.obs), rows, Do, 64, 1, K, N.ptr(L.actor), N.ptr(b.avail), N.ptr(L.logits), s), "f"))
k["advantage"] = timed(lambda: N.check(lib.cm_coma_advantage(N.ptr(L.logits), N.ptr(L.q), N.ptr(b.action), N.ptr(b.ep_len), E, A, T, K, N.ptr(b.adv),
                                                              N.ptr(L.tstats), N.ptr(L.ws), L.ws.numel(), s), "a"))
k["coma_actor_fwd_bwd"] = timed(lambda: N.check(lib.cm_coma_actor_fwd_bwd(N.ptr(b.obs), N.ptr(b.avail), N.ptr(b.action), N.ptr(b.adv), N.ptr(b.ep_len), E, A, T,
                                                                           Do, 64, 1, K, N.ptr(L.actor), 1e-3, N.ptr(L.g_actor), N.ptr(L.ws), L.ws.numel(), s), "p"))
Pc = Dc * Hc + Hc * Hc + Hc * K
flop_ref = rows * (2 * Pc + 2 * Pc + 2 * (Pc - Dc * Hc))  # the reference's (materialised-input) critic: fwd + dW + dX per row
# FLOPs the factored schedule actually needs: obs block per row, state block per (e,t), action block as one-hot GEMM in the backward only
Pf = Do * Hc + Hc * Hc + Hc * K
flop = rows * (4 * Pf + 2 * (Pf - Do * Hc)) + E * T * 4 * Ds * Hc + rows * 2 * (A - 1) * K * Hc
ach = flop / (k["critic_fwd_bwd"] * 1e-3) / 1e12
out = {"metric": "env-steps/sec (agents x envs x steps), COMA full iteration", "value": E * A * T * args.steps / dt, "unit": "agent-env-steps/s",
       "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True, "dtype": "f32",
       "data": "synthetic", "config": {"workload": f"COMA synthetic-MPE {E} envs x {A} agents x {T} steps, actor 2x64, critic input {Dc} -> 2x{Hc} -> {K}"},
       "phase_ms": {"rollout_eps_mixed": ph[0], "targets": ph[1], "update": ph[2]}, "kernel_ms": k,
       "roofline": {"kernel": "cm_coma_critic_fwd_bwd (k_wide_gemm<EPI_COMA> (S + z0 in one launch) + k_mlp<1,M_QCRITIC> + k_coma_bwd_gather + k_dw0_stream)" if Hc <= 64 else "cm_coma_critic_fwd_bwd (k_wide_gemm<EPI_COMA> (S + z0 in one launch) + k_mlp128<M_QCRITIC> + k_coma_bwd_gather + k_dw0_stream)", "bound": "mfma", "achieved": ach,
                    "peak": 157.3, "unit": "TFLOP/s", "frac": ach / 157.3, "flop_per_launch": flop, "reference_schedule_flop": flop_ref,
                    "reference_schedule_equiv_tflops": flop_ref / (k["critic_fwd_bwd"] * 1e-3) / 1e12,
                    "algorithmic_bytes_per_launch": rows * (4 * Do + 12) + E * T * 4 * Ds}}
if not args.no_cpu_baseline:
    from oracle import coma as C  # baseline only
    from oracle import restatement as R
    Ec = args.cpu_envs
    g = torch.Generator().manual_seed(0)
    batch = dict(obs=torch.randn(Ec, T, A, Do, generator=g), states=torch.randn(Ec, T, Ds, generator=g), avail=torch.ones(Ec, T, A, K, dtype=torch.bool),
                 actions=torch.randint(0, K, (Ec, T, A), generator=g), reward=torch.randn(Ec, T, generator=g), mask=torch.ones(Ec, T, dtype=torch.bool))
    apl, cpl = init_params_like_torch(aspec), init_params_like_torch(cspec)
    hp = dict(gamma=0.99, td_lambda=0.8, normalize_advantage=1.0, normalize_return=0.0, target_network_update_freq=1.0, polyak=0.005, entropy_coef=1e-3,
              use_tdlamda=1.0, nsteps=1.0, clip_gradients=-1.0, optimizer="Adam", learning_rate_actor=5e-4, learning_rate_critic=5e-4)
    t1 = time.perf_counter()
    C.update(apl, cpl, [p.clone() for p in cpl], batch, hp)
    dtc = time.perf_counter() - t1
    out["cpu_baseline"] = {"value": Ec * A * T / dtc, "unit": "agent-env-steps/s (learner only, no rollout)", "cores": torch.get_num_threads(), "kind": "port",
                           "sample": f"one oracle/coma.py update (batched torch-CPU restatement) at {Ec} envs x {A} agents x {T} steps: {dtc:.2f} s"}
print(json.dumps(out))
