#!/usr/bin/env python3
"""Why does the 1024-env share time 3.5 ms inside bench.py's share leg and 2.7 ms as `bench.py --envs 1024`?"""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
dev = torch.device("cuda:0")
def t(E, steps, warm, tag):
    w = bench.Workload("cfg3", E, 0, dev)
    r = w.run(steps, warm)
    print(f"{tag}: E={E} steps={steps} warm={warm}: {r['ms_per_step']:.3f} ms  {r['phase_ms']}", flush=True)
    w.close()
t(1024, 20, 6, "fresh process")
t(1024, 40, 10, "again")
t(4096, 5, 2, "full size")
t(2048, 20, 6, "after full size")
t(1024, 20, 6, "after full size")
t(1024, 20, 6, "again")
os.environ["CM_CRITIC_SCHEDULE"] = "split"
t(1024, 20, 6, "split schedule")
