"""A/B of the two tilings of the fused rollout (cm_rollout_spread): 64-row tiles vs 16-row tiles (CM_ROLLOUT_TILE).
Prints whether the two produce bit-identical rollouts and their time per launch at several env counts."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from cleanmarl_amd.learner import NetSpec, flatten_params, init_params_like_torch  # noqa: E402
from cleanmarl_amd.rollout import SyntheticSpreadRollout  # noqa: E402

dev = torch.device("cuda:0")


def run(E, A, T, tile, reps=0):
    os.environ["CM_ROLLOUT_TILE"] = str(tile)
    torch.manual_seed(3)
    r = SyntheticSpreadRollout(E, A, T, seed=7, device=dev, env_offset=5)
    spec = NetSpec(r.Do, 64, 1, 5)
    p = flatten_params(init_params_like_torch(spec), dev)
    b = r.collect(p, spec, fused=True)
    torch.cuda.synchronize()
    out = {k: getattr(b, k).clone() for k in ("obs", "state", "action", "logp", "reward")}
    ms = None
    if reps:
        for _ in range(5):
            r.collect(p, spec, fused=True)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            r.collect(p, spec, fused=True)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
    return out, ms


for (E, A, T) in ([] if os.environ.get("RO16_ONLY_TIME") else [(37, 8, 12), (50, 3, 9), (10, 5, 6), (512, 8, 128)]):
    a, _ = run(E, A, T, 64)
    b, _ = run(E, A, T, 16)
    same = {k: bool(torch.equal(a[k], b[k])) for k in a}
    frac = (a["action"] == b["action"]).float().mean().item()
    print(f"E={E} A={A} T={T}: bit-identical {same}  actions equal {frac:.6f}  max|dlogp| {(a['logp'] - b['logp']).abs().max().item():.3e}")
for (E, A) in [(512, 8), (1024, 8), (2048, 8), (4096, 8), (1024, 3), (128, 3), (1024, 5)]:
    t64 = run(E, A, 128, 64, 30)[1]
    t16 = run(E, A, 128, 16, 30)[1]
    print(f"E={E:5d} A={A} T=128: 64-row {t64:.3f} ms   16-row {t16:.3f} ms")
