#!/usr/bin/env python3
"""Outcome of 300 full-size MAPPO iterations through the CLI for several seeds and both product forms of the actor pass (CM_MLP_FORMS=hand: the
instantiation with the 4x4x1 head, the default at this size; loop: the compiler-scheduled twin with the 16x16x4 head): is a different end state a
property of the arithmetic or of the task (chaotic divergence of trajectories that agree to 1e-7 per update)?"""
import math
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CODE = r'''
import sys, math
sys.path.insert(0, %r)
from cleanmarl_amd.driver import run
E, T, A, iters, seed = 4096, 128, 8, 300, int(sys.argv[1])
out = run("mappo_multienvs", ["--env_type=synthetic", f"--batch_size={E}", f"--synthetic_agents={A}", f"--synthetic_steps={T}", f"--total_timesteps={E * T * iters}",
                              "--eval_steps=50", "--num_eval_ep=10", "--log_every=10", "--actor_hidden_dim=64", "--critic_hidden_dim=64", f"--seed={seed}"])
h = out["history"]
s = lambda tag: [v for t, v, _ in h if t == tag]
en, cl, rw = s("train/entropy"), s("train/critic_loss"), s("rollout/ep_reward")
mid = len(rw) // 2
print(f"entropy {en[0]:.4f} -> {en[-1]:.4f}  critic_loss {cl[0]:.1f} -> {cl[-1]:.1f}  ep_reward {rw[0]:.1f} -> {rw[mid]:.1f} -> {rw[-1]:.1f}  finite {all(math.isfinite(x) for x in en + cl + rw)}")
''' % ROOT
for seed in (1, 2, 3, 4):
    for forms in ("hand", "loop"):
        env = dict(os.environ, CM_MLP_FORMS=forms)
        r = subprocess.run([sys.executable, "-c", CODE, str(seed)], env=env, capture_output=True, text=True)
        line = [ln for ln in r.stdout.splitlines() if ln.startswith("entropy")]
        print(f"seed {seed} forms {forms}: {line[-1] if line else r.stderr[-300:]}", flush=True)
