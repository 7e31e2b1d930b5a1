#!/usr/bin/env python3
"""Steady-state ms per iteration of the PRODUCT surface -- cleanmarl_amd.driver.run, i.e. the mappo_multienvs CLI with its logging
cadence -- beside bench.py's inner-loop number (bench.py times the inner loop, the product is driver.run).
Two runs per configuration with N and 2N iterations; their difference / N cancels start-up (library load, first-use allocations,
the first iterations' code-object loads)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cleanmarl_amd.driver import run  # noqa: E402


def timed(E, A, T, iters, log_every, eval_flags=("--eval_steps=1000000000",)):
    t0 = time.perf_counter()
    out = run("mappo_multienvs", ["--env_type=synthetic", f"--batch_size={E}", f"--synthetic_agents={A}", f"--synthetic_steps={T}",
                                  f"--total_timesteps={E * T * iters}", f"--log_every={log_every}",
                                  "--actor_hidden_dim=64", "--critic_hidden_dim=64"] + list(eval_flags))
    import torch
    torch.cuda.synchronize()
    assert out["training_step"] == 3 * iters, out["training_step"]
    return time.perf_counter() - t0


if __name__ == "__main__":
    os.chdir("/tmp")
    timed(64, 3, 16, 20, 10 ** 9)  # library load, first-use allocations, code objects: not part of any measured run
    for (E, A, T, n) in ((4096, 8, 128, 200), (512, 8, 128, 1000), (1024, 3, 128, 1000)):
        for log_every, label in ((10 ** 9, "no per-iteration logging"), (1, "--log_every=1 (scalars fetched every iteration)")):
            t1, t2 = timed(E, A, T, n, log_every), timed(E, A, T, 2 * n, log_every)
            print(f"driver.run {E} envs x {A} agents x {T} steps, {label}: {1e3 * (t2 - t1) / n:.3f} ms per iteration "
                  f"(runs of {n} / {2 * n} iterations: {t1:.2f} s / {t2:.2f} s)", flush=True)
        # the reference's DEFAULT evaluation cadence (cleanmarl/mappo_multienvs.py:66-69: eval_steps 50, num_eval_ep 10; actions sampled)
        # and the build's --greedy_eval, against the eval-off rate above: evaluation is one batched device rollout on its own
        # low-priority stream (cleanmarl_amd/evaluate.py), so the ratio should be ~1
        base = None
        for flags, label in ((("--eval_steps=1000000000",), "eval off"), ((), "reference defaults --eval_steps=50 --num_eval_ep=10"),
                             (("--greedy_eval",), "defaults + --greedy_eval"), (("--eval_steps=5",), "--eval_steps=5 (10 x the default cadence)")):
            t1, t2 = timed(E, A, T, n, 10, flags), timed(E, A, T, 2 * n, 10, flags)
            ms = 1e3 * (t2 - t1) / n
            base = ms if base is None else base
            print(f"driver.run {E} envs x {A} agents x {T} steps, --log_every=10, {label}: {ms:.3f} ms per iteration = {ms / base:.3f} x eval-off",
                  flush=True)
