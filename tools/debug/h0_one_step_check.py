#!/usr/bin/env python3
"""One full iteration (rollout, targets, update) of a bench workload from identical initial state with CM_CRITIC_H0=0 and =1: the critic's parameters after
the update must agree to rounding (the first epoch's gradient comes from the value pass's activations in one run, from its own forward half in the other),
the actor's bit for bit.    python tools/debug/h0_one_step_check.py [cfg3|cfg4|cfg5]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "cfg5"
dev = torch.device("cuda:0")
out = {}
for h in ("0", "1"):
    os.environ["CM_CRITIC_H0"] = h
    torch.manual_seed(0)
    w = bench.Workload(name, bench.WORKLOADS[name][0], 0, dev)
    w.one_step()
    torch.cuda.synchronize()
    L = w.learner
    out[h] = (L.actor.clone(), L.critic_params().clone(), L._h0_key is not None or L._h0 is not None)
a0, c0, k0 = out["0"]
a1, c1, k1 = out["1"]
d = (c0 - c1).abs().max().item()
print(f"{name}: hand-off active {k0} / {k1}; actor equal {torch.equal(a0, a1)}; critic max |diff| {d:.3e} (max |param| {c0.abs().max().item():.3f}, "
      f"max |update| {(c0 - c0.new_tensor(0)).abs().max().item():.3f})")
assert k1 and not k0 and d < 1e-5
