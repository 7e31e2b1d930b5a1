#!/usr/bin/env python3
"""Time the learner (targets + epochs of update) of every BASELINE.json config on random device-resident
buffers of that shape (secondary configs are parity-test cases; this records their update latency).
usage: python tools/bench_configs.py > profiles/rNN_configs.txt"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cleanmarl_amd.gru import GRUPPOLearner  # noqa: E402
from cleanmarl_amd.learner import DeviceBatch, HParams, NetSpec, PPOLearner  # noqa: E402

CFG = [  # name, algo, gru, E, A, T, Do, Ds, K
    ("cfg2 MAPPO 1024x3x128", "mappo", False, 1024, 3, 128, 21, 54, 5),
    ("cfg3 MAPPO 4096x8x128", "mappo", False, 4096, 8, 128, 56, 384, 5),
    ("cfg4 IPPO 2048x10x256 (1 GPU's share of 8: 256 envs)", "ippo", False, 256, 10, 256, 115, 243, 17),
    ("cfg4 IPPO 2048x10x256 (all envs on 1 GPU)", "ippo", False, 2048, 10, 256, 115, 243, 17),
    ("cfg5 MAPPO-GRU 1024x5x128 tbptt=10", "mappo", True, 1024, 5, 128, 35, 150, 5),
]
if os.environ.get("CM_BENCH_ALIGNED"):  # what 16-byte aligned rows would buy: the same configs with feature widths rounded up to 4
    CFG = [(n + " [widths rounded up to 4]", al, g, E, A, T, (Do + 3) // 4 * 4, (Ds + 3) // 4 * 4, K) for (n, al, g, E, A, T, Do, Ds, K) in CFG]
dev = torch.device("cuda:0")
for name, algo, gru, E, A, T, Do, Ds, K in CFG:
    g = torch.Generator().manual_seed(0)
    b = DeviceBatch(E, A, T, Do, Ds, K, dev)
    b.obs.normal_(); b.state.normal_(); b.reward.normal_()
    av = (torch.rand(E, A, T, K, generator=g) < 0.7); av[..., 0] = True
    b.avail.copy_(av.to(torch.uint8))
    b.action.zero_(); b.logp.fill_(-1.5); b.ep_len.fill_(T)
    torch.manual_seed(1)
    aspec = NetSpec(Do, 64, 0 if gru else 1, K, "gru" if gru else "mlp")
    cspec = NetSpec(Ds if algo == "mappo" else Do, 64, 1, 1)
    L = (GRUPPOLearner if gru else PPOLearner)(algo, aspec, cspec, A, HParams(), dev)
    for _ in range(2):
        L.train_iteration(b)
    torch.cuda.synchronize()
    n = 5
    t0 = time.perf_counter()
    for _ in range(n):
        L.train_iteration(b)
    torch.cuda.synchronize()
    ms = 1e3 * (time.perf_counter() - t0) / n
    print(f"{name:60s} targets+3 epochs: {ms:8.2f} ms  ({ms / 3:6.2f} ms/epoch)  {E * A * T / ms / 1e3:8.1f} M agent-steps/s (learner only)")
