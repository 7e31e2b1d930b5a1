#!/usr/bin/env python3
"""Does the queue mapping / priority of the two streams change how soon main-stream kernels start beside the critic's?
    python tools/probes/stream_priority.py [envs]   -> ms per step for (main stream, side stream) priority pairs"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from cleanmarl_amd import _native as N  # noqa: E402

E = int(sys.argv[1]) if len(sys.argv) > 1 else 512
K = 60
dev = torch.device("cuda:0")
orig = N.low_priority_stream
for main_p, side_p in (("default", "low"), ("default", "normal"), ("high", "low"), ("high", "normal"), ("normal", "low"), ("normal", "normal")):
    N.low_priority_stream = orig if side_p == "low" else (lambda device: torch.cuda.Stream(device=device))
    main = None if main_p == "default" else torch.cuda.Stream(device=dev, priority=-1 if main_p == "high" else 0)
    ctx = torch.cuda.stream(main) if main is not None else torch.cuda.stream(torch.cuda.current_stream())
    with ctx:
        w = bench.Workload("cfg3", E, 0, dev)
        for _ in range(10):
            w.one_step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(K):
            w.one_step()
        torch.cuda.synchronize()
        print(f"envs {E}: main={main_p:8s} side={side_p:7s} {1e3 * (time.perf_counter() - t0) / K:.3f} ms/step", flush=True)
    del w
