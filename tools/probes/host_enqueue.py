#!/usr/bin/env python3
"""Is a small share host-bound?  Times K bench steps twice: the host's enqueue time (before the final synchronise) and the wall time.
    python tools/probes/host_enqueue.py [envs] [steps]      (cProfile of the enqueue loop with CM_PROBE_PROFILE=1)"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

E = int(sys.argv[1]) if len(sys.argv) > 1 else 512
K = int(sys.argv[2]) if len(sys.argv) > 2 else 100
dev = torch.device("cuda:0")
w = bench.Workload("cfg3", E, 0, dev)
for _ in range(10):
    w.one_step()
torch.cuda.synchronize()
if os.environ.get("CM_PROBE_PROFILE"):
    import cProfile
    import pstats
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(K):
        w.one_step()
    pr.disable()
    torch.cuda.synchronize()
    pstats.Stats(pr).sort_stats("cumulative").print_stats(35)
    sys.exit(0)
t0 = time.perf_counter()
for _ in range(K):
    w.one_step()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"envs {E}: host enqueue {1e3 * (t1 - t0) / K:.3f} ms/step, wall {1e3 * (t2 - t0) / K:.3f} ms/step")
