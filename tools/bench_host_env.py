#!/usr/bin/env python3
"""Host-env plumbing throughput: reference-style pipe-per-env protocol vs the shared-memory
batched-step vector env, same CPU synthetic env, actor replaced by a trivial numpy policy so only the
vectorisation is measured.   usage: python tools/bench_host_env.py [E] [A] [T]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cleanmarl_amd.driver import host_rollout, host_rollout_shm  # noqa: E402
from cleanmarl_amd.env.shm_vector import ShmVectorEnv  # noqa: E402
from cleanmarl_amd.env.vector import PipeVectorEnv  # noqa: E402

E, A, T = (int(x) for x in (sys.argv[1:4] + ["256", "8", "128"][len(sys.argv) - 1:]))
fac = dict(env_type="synthetic_cpu", env_name="x", env_family="mpe", agent_ids=True, kwargs={}, seed=1, synthetic=dict(agents=A, steps=T))


class Policy:
    def act(self, obs, avail, h=None, seed=0, eps=0.0):
        n = obs.shape[0] * obs.shape[1]
        return np.zeros(n, np.int32), np.zeros(n, np.float32), None


for name, make, roll in (("pipe-per-env (reference protocol)", lambda: PipeVectorEnv(E, fac), host_rollout),
                         ("shared-memory batched step", lambda: ShmVectorEnv(E, fac), host_rollout_shm)):
    v = make()
    roll(v, Policy(), E, A, 0, False, torch.device("cpu"))  # warm-up episode
    t0 = time.perf_counter()
    roll(v, Policy(), E, A, 0, False, torch.device("cpu"))
    dt = time.perf_counter() - t0
    v.close()
    print(f"{name:36s} E={E} A={A} T={T}: {dt:7.2f} s per rollout+collate = {E * A * T / dt / 1e3:8.1f} k agent-env-steps/s ({os.cpu_count()} host cores)")

if torch.cuda.is_available():  # the same two shared-memory paths with the REAL actor on the GPU: host-staged steps vs pinned blocks
    from cleanmarl_amd.driver import HostActor
    from cleanmarl_amd.host_rollout import PinnedHostRollout
    from cleanmarl_amd.learner import HParams, NetSpec, PPOLearner
    dev = torch.device("cuda:0")
    L = PPOLearner("mappo", NetSpec(7 * A, 64, 1, 5), NetSpec(6 * A * A, 64, 1, 1), A, HParams(), dev)
    v = ShmVectorEnv(E, fac)
    ha = HostActor(L, A, False, dev)
    host_rollout_shm(v, ha, E, A, 1, False, dev)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    host_rollout_shm(v, ha, E, A, 2, False, dev)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"{'shm + HIP actor, host-staged steps':36s} E={E} A={A} T={T}: {dt:7.2f} s per rollout+collate = {E * A * T / dt / 1e3:8.1f} k agent-env-steps/s")
    pr = PinnedHostRollout(v, L, False, dev, t_cap=T)
    pr.collect(1)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    pr.collect(2)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"{'shm + HIP actor, pinned blocks':36s} E={E} A={A} T={T}: {dt:7.2f} s per rollout (batch built on device) = {E * A * T / dt / 1e3:8.1f} k agent-env-steps/s")
    pr.close(); v.close()
