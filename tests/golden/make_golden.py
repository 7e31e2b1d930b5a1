#!/usr/bin/env python3
"""Generate golden fixtures by running the UNMODIFIED reference scripts.

Runs ONLY in the build container (needs /root/reference).  Nothing here travels
to the GPU box except the .npz files this script writes next to itself.

Method: the reference learner lives inside the
``if __name__ == "__main__"`` block of e.g. cleanmarl/mappo_multienvs.py, so we
execute the file with ``runpy.run_path(..., run_name="__main__")`` after
injecting stub modules for the packages that are not installed here
(tyro, tensorboard, env.* wrappers).  ``run_path`` returns the script's globals,
from which we read the collated batch, the TD(lambda) returns / advantages and
the per-epoch logged scalars; initial weights, per-step gradients and post-step
weights are captured by subclassing the torch optimiser the script looks up with
``getattr(optim, args.optimizer)``.

The synthetic environment (tests/stub_envs.py::SynthEnv) is OURS (deterministic, numpy, implements
the reference's CommonInterface surface: cleanmarl/env/common_interface.py:5-23); it lives next to the other
test stand-ins so that the host-collation tests can replay the very episodes the goldens were captured on.

usage:  python tests/golden/make_golden.py            # writes tests/golden/*.npz
"""
import dataclasses
import os
import runpy
import sys
import types

import numpy as np
import torch
import torch.optim as optim

REF = "/root/reference/cleanmarl"
OUT = os.path.dirname(os.path.abspath(__file__))


sys.path.insert(0, os.path.dirname(OUT))
from stub_envs import SynthEnv  # noqa: E402  the deterministic env the goldens were collected on (OURS; tests replay it: test_pins.py)


class _SW:
    """stand-in for torch.utils.tensorboard.SummaryWriter"""
    inst = None

    def __init__(self, *a, **k):
        self.log = []
        _SW.inst = self

    def add_scalar(self, tag, v, step):
        self.log.append((tag, float(v), int(step)))

    def add_text(self, *a, **k):
        pass

    def close(self):
        pass


def _install_stubs(overrides):
    tb = types.ModuleType("torch.utils.tensorboard")
    tb.SummaryWriter = _SW
    sys.modules["torch.utils.tensorboard"] = tb
    ty = types.ModuleType("tyro")
    ty.cli = lambda cls: dataclasses.replace(cls(), **overrides)
    sys.modules["tyro"] = ty
    pkg = types.ModuleType("env")
    pkg.__path__ = []
    sys.modules["env"] = pkg
    for mod, cls in [("pettingzoo_wrapper", "PettingZooWrapper"),
                     ("smaclite_wrapper", "SMACliteWrapper"), ("lbf", "LBFWrapper")]:
        m = types.ModuleType("env." + mod)
        setattr(m, cls, lambda family=None, env_name=None, map_name=None, agent_ids=True, **k: SynthEnv(agent_ids))
        sys.modules["env." + mod] = m


def run_reference(script, overrides, env_spec):
    """Execute one reference script for exactly one training iteration."""
    SynthEnv.counter = 0
    SynthEnv.spec = dict(SynthEnv.spec, **env_spec)
    ov = dict(env_type="pz", total_timesteps=1, eval_steps=10 ** 9, seed=1)
    ov.update(overrides)
    _install_stubs(ov)

    snaps = []  # one dict per optimiser, in construction order (actor, critic)

    def make(base):
        class Snap(base):
            def __init__(self, params, **k):
                params = list(params)
                self._snap = dict(init=[p.detach().clone().numpy() for p in params],
                                  grads=[], after=[])
                snaps.append(self._snap)
                super().__init__(params, **k)

            def step(self, *a, **k):
                ps = [p for g in self.param_groups for p in g["params"]]
                self._snap["grads"].append([p.grad.detach().clone().numpy() for p in ps])
                r = super().step(*a, **k)
                self._snap["after"].append([p.detach().clone().numpy() for p in ps])
                return r
        Snap.__name__ = base.__name__
        return Snap

    saved = {n: getattr(optim, n) for n in ("Adam", "AdamW", "SGD", "RMSprop")}
    try:
        for n, b in saved.items():
            setattr(optim, n, make(b))
        g = runpy.run_path(os.path.join(REF, script), run_name="__main__")
    finally:
        for n, b in saved.items():
            setattr(optim, n, b)
    return g, snaps, list(_SW.inst.log)


def _flat(lst):
    return np.concatenate([np.asarray(x, dtype=np.float32).reshape(-1) for x in lst])


def pack(g, snaps, log, extra=None):
    a = g["args"]
    out = {}
    for k in ("b_obs", "b_actions", "b_log_probs", "b_reward", "b_states",
              "b_avail_actions", "b_done", "b_mask", "return_lambda", "advantages"):
        out[k] = g[k].detach().cpu().numpy()
    for name, s in zip(("actor", "critic"), snaps):
        out[f"{name}_shapes"] = np.array([len(p.shape) and p.shape[0] for p in s["init"]] , dtype=np.int64)
        for i, p in enumerate(s["init"]):
            out[f"{name}_init_{i}"] = p
        out[f"{name}_nparam"] = np.int64(len(s["init"]))
        out[f"{name}_grads"] = np.stack([_flat(x) for x in s["grads"]])
        out[f"{name}_after"] = np.stack([_flat(x) for x in s["after"]])
    for k in ("actor_losses", "critic_losses", "entropies_bonuses", "kl_divergences",
              "actor_gradients", "critic_gradients", "clipped_ratios"):
        out[k] = np.array([float(x) for x in g[k]], dtype=np.float64)
    hp = {f.name: getattr(a, f.name) for f in dataclasses.fields(a)}
    for k, v in hp.items():
        if isinstance(v, (bool, int, float)):
            out["hp_" + k] = np.float64(v)
        else:
            out["hp_" + k] = np.array(str(v))
    out["log_tags"] = np.array([t for t, _, _ in log])
    out["log_vals"] = np.array([v for _, v, _ in log], dtype=np.float64)
    out["log_steps"] = np.array([s for _, _, s in log], dtype=np.int64)
    if extra:
        out.update(extra)
    return out


def pack_coma(g, snaps, log):
    """COMA (cleanmarl/coma_multienvs.py): one iteration = TD(lambda)/n-step targets from the TARGET critic, ONE critic
    step, polyak target update, ONE actor step (no epochs, no stored log-probs)."""
    a = g["args"]
    out = {}
    for k in ("b_obs", "b_actions", "b_reward", "b_states", "b_avail_actions", "b_done", "b_mask", "return_lambda"):
        out[k] = g[k].detach().cpu().numpy()
    for name, sn in zip(("actor", "critic"), snaps):
        for i, p in enumerate(sn["init"]):
            out[f"{name}_init_{i}"] = p
        out[f"{name}_nparam"] = np.int64(len(sn["init"]))
        out[f"{name}_grads"] = np.stack([_flat(x) for x in sn["grads"]])
        out[f"{name}_after"] = np.stack([_flat(x) for x in sn["after"]])
    out["target_after"] = _flat([p.detach().numpy() for p in g["target_critic"].parameters()])
    out["epsilon"] = np.float64(g["epsilon"])
    for k in ("cr_loss", "ac_loss", "entropies", "actor_gradients", "critic_gradients"):
        out[k] = np.float64(float(g[k]))
    hp = {f.name: getattr(a, f.name) for f in dataclasses.fields(a)}
    for k, v in hp.items():
        out["hp_" + k] = np.float64(v) if isinstance(v, (bool, int, float)) else np.array(str(v))
    out["log_tags"] = np.array([t for t, _, _ in log])
    out["log_vals"] = np.array([v for _, v, _ in log], dtype=np.float64)
    out["log_steps"] = np.array([st for _, _, st in log], dtype=np.int64)
    return out


CASES = {
    # name: (script, overrides, env_spec)
    "mappo_dense": ("mappo_multienvs.py",
                    dict(batch_size=8),
                    dict(A=3, obs_raw=6, K=5, horizon=16, ragged=False, avail_p=1.0, state_dim=None, done_mode="truncate")),
    "mappo_ragged_norm": ("mappo_multienvs.py",
                          dict(batch_size=7, actor_hidden_dim=64, normalize_reward=True, normalize_advantage=True,
                               normalize_return=True, clip_gradients=0.5),
                          dict(A=3, obs_raw=6, K=5, horizon=18, ragged=True, avail_p=0.7, state_dim=None, done_mode="done")),
    "mappo_deep": ("mappo_multienvs.py",
                   dict(batch_size=5, actor_hidden_dim=64, actor_num_layers=2, critic_hidden_dim=32,
                        critic_num_layers=0, clip_gradients=10.0, optimizer="AdamW"),
                   dict(A=2, obs_raw=35, K=4, horizon=12, ragged=True, avail_p=1.0, state_dim=None, done_mode="truncate")),
    # widths / depths beyond the fused 64-wide kernels -> the layered schedule (csrc/cm_mlp_wide.h)
    "mappo_wide": ("mappo_multienvs.py",
                   dict(batch_size=6, actor_hidden_dim=128, critic_hidden_dim=128, critic_num_layers=3, normalize_advantage=True,
                        clip_gradients=0.5),
                   dict(A=3, obs_raw=6, K=5, horizon=14, ragged=True, avail_p=0.7, state_dim=None, done_mode="done")),
    # --optimizer is looked up with getattr(optim, ...): RMSprop (the COMA paper's choice) and plain SGD with torch's defaults
    "mappo_rmsprop": ("mappo_multienvs.py",
                      dict(batch_size=5, optimizer="RMSprop", clip_gradients=0.5, learning_rate_actor=5e-4, learning_rate_critic=5e-4),
                      dict(A=3, obs_raw=6, K=5, horizon=12, ragged=True, avail_p=0.8, state_dim=None, done_mode="done")),
    "ippo_sgd": ("ippo_multienvs.py",
                 dict(batch_size=5, optimizer="SGD", learning_rate_actor=1e-2, learning_rate_critic=1e-2),
                 dict(A=2, obs_raw=7, K=4, horizon=10, ragged=False, avail_p=1.0, state_dim=9, done_mode="truncate")),
    "ippo_dense": ("ippo_multienvs.py",
                   dict(batch_size=8, critic_hidden_dim=64, actor_hidden_dim=64),
                   dict(A=4, obs_raw=10, K=6, horizon=14, ragged=False, avail_p=1.0, state_dim=17, done_mode="truncate")),
    "ippo_ragged_norm": ("ippo_multienvs.py",
                         dict(batch_size=6, normalize_reward=True, normalize_advantage=True, normalize_return=True,
                              clip_gradients=0.5),
                         dict(A=3, obs_raw=11, K=7, horizon=20, ragged=True, avail_p=0.6, state_dim=19, done_mode="done")),
    "mappo_lstm_ragged": ("mappo_lstm_multienvs.py",
                          dict(batch_size=6, actor_hidden_dim=32, tbptt=4, epochs=2, clip_gradients=0.5),
                          dict(A=3, obs_raw=6, K=5, horizon=21, ragged=True, avail_p=0.8, state_dim=None, done_mode="done")),
    "mappo_lstm_dense": ("mappo_lstm_multienvs.py",
                         dict(batch_size=4, actor_hidden_dim=64, tbptt=10, epochs=3),
                         dict(A=4, obs_raw=10, K=5, horizon=24, ragged=False, avail_p=1.0, state_dim=None, done_mode="truncate")),
    "ippo_lstm_ragged": ("ippo_lstm_multienvs.py",
                         dict(batch_size=5, actor_hidden_dim=32, critic_hidden_dim=32, epochs=2,
                              normalize_advantage=True),
                         dict(A=3, obs_raw=7, K=6, horizon=17, ragged=True, avail_p=0.7, state_dim=13, done_mode="done")),
    # COMA: TD(lambda) targets (default) and n-step targets; per-time-step advantage normalisation is the default
    "coma_tdlambda": ("coma_multienvs.py",
                      dict(batch_size=6, actor_hidden_dim=64, critic_hidden_dim=64, clip_gradients=0.5, normalize_return=True),
                      dict(A=3, obs_raw=6, K=5, horizon=15, ragged=True, avail_p=0.7, state_dim=None, done_mode="done")),
    # the reference's DEFAULT critic width (128): layered schedule on the materialised critic input
    "coma_default_width": ("coma_multienvs.py",
                           dict(batch_size=5, clip_gradients=0.5),
                           dict(A=3, obs_raw=6, K=5, horizon=13, ragged=True, avail_p=0.8, state_dim=None, done_mode="done")),
    "coma_nstep": ("coma_multienvs.py",
                   dict(batch_size=5, actor_hidden_dim=32, critic_hidden_dim=64, use_tdlamda=False, nsteps=3,
                        normalize_advantage=False, normalize_reward=True),
                   dict(A=4, obs_raw=9, K=6, horizon=12, ragged=True, avail_p=1.0, state_dim=14, done_mode="truncate")),
}


def main(names=None):
    torch.set_num_threads(1)  # deterministic summation order in the reference run
    for name, (script, ov, spec) in CASES.items():
        if names and name not in names:
            continue
        g, snaps, log = run_reference(script, ov, spec)
        if script.startswith("coma"):
            out = pack_coma(g, snaps, log)
            if ov.get("normalize_reward"):
                g2, _, _ = run_reference(script, dict(ov, normalize_reward=False), spec)
                assert np.array_equal(g2["b_actions"].numpy(), g["b_actions"].numpy())
                out["b_reward_raw"] = g2["b_reward"].numpy()
            path = os.path.join(OUT, name + ".npz")
            np.savez_compressed(path, **out)
            print(f"{name}: wrote {path} ({os.path.getsize(path)/1024:.0f} KiB) B,T,A={out['b_obs'].shape[:3]}")
            continue
        extra = {}
        if ov.get("normalize_reward"):
            # identical run without reward normalisation -> raw rewards (rollout is
            # a pure function of the torch seed; normalisation happens after it)
            ov2 = dict(ov, normalize_reward=False)
            g2, _, _ = run_reference(script, ov2, spec)
            assert np.array_equal(g2["b_actions"].numpy(), g["b_actions"].numpy())
            extra["b_reward_raw"] = g2["b_reward"].numpy()
        out = pack(g, snaps, log, extra)
        path = os.path.join(OUT, name + ".npz")
        np.savez_compressed(path, **out)
        print(f"{name}: wrote {path} ({os.path.getsize(path)/1024:.0f} KiB) "
              f"B,T,A={out['b_obs'].shape[:3]} actor_steps={len(out['actor_grads'])} critic_steps={len(out['critic_grads'])}")


if __name__ == "__main__":
    main(sys.argv[1:] or None)
