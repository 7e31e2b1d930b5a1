"""Where do the largest gradient differences against the oracle sit?  (round 6: the scale-aware bars of tests/parity.py flagged three
cases at 1.7e-4 .. 6e-4 of the largest gradient entry, all with 17 actions and availability masks.)  Prints, per parameter block, the
largest |HIP - oracle| relative to the largest oracle entry of the WHOLE gradient, and for the worst entries the values themselves."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_hip_parity as t  # noqa: E402
from oracle import restatement as R  # noqa: E402


def blocks(spec):
    out, o = [], 0
    for i, sh in enumerate(spec.shapes()):
        n = int(np.prod(sh)); out.append((i, sh, o, o + n)); o += n
    return out


def report(name, spec, got, ref):
    got, ref = np.asarray(got, np.float64), np.asarray(ref, np.float64)
    mx = np.abs(ref).max()
    print(f"  {name}: max|ref| {mx:.3e}  max|diff|/max|ref| {np.abs(got - ref).max() / mx:.3e}")
    for i, sh, a, b in blocks(spec):
        d = np.abs(got[a:b] - ref[a:b])
        k = int(d.argmax())
        print(f"    block {i} {sh}: {d.max() / mx:.3e}   worst entry {np.unravel_index(k, sh)}: hip {got[a + k]:+.6e} ref {ref[a + k]:+.6e}")


def mlp_case(algo, E, A, T, Do, Ds, K, H, L):
    from cleanmarl_amd.learner import DeviceBatch, HParams, NetSpec, PPOLearner, init_params_like_torch
    torch.manual_seed(1)
    batch = t._random_case(123, E, A, T, Do, Ds, K)
    aspec, cspec = NetSpec(Do, H, L, K), NetSpec(Ds if algo == "mappo" else Do, H, L, 1)
    ap, cp = init_params_like_torch(aspec), init_params_like_torch(cspec)
    hp = dict(gamma=0.99, td_lambda=0.95, normalize_advantage=True, normalize_return=False, epochs=2, ppo_clip=0.2, entropy_coef=0.01,
              clip_gradients=0.5, optimizer="Adam", learning_rate_actor=8e-4, learning_rate_critic=8e-4)
    dev = torch.device("cuda:0")
    b = DeviceBatch.from_reference_layout(batch["obs"], batch["actions"], batch["log_probs"], batch["reward"], batch["states"], batch["avail"], batch["mask"], dev)
    Lr = PPOLearner(algo, aspec, cspec, A, HParams(**hp), dev, actor_params=[p.clone() for p in ap], critic_params=[p.clone() for p in cp])
    recs = Lr.train_iteration(b, keep_grads=True)
    ret, adv, orecs = R.mlp_update(ap, cp, batch, hp, algo)
    print(f"MLP {algo} E{E} A{A} T{T} Do{Do} K{K} H{H} L{L}")
    for e, (r, o) in enumerate(zip(recs, orecs)):
        report(f"epoch {e} actor", aspec, r["actor_grads"].cpu().numpy(), R.flat(o["actor_grads"]).numpy())
    # how close are rows to the points where the clipped surrogate's derivative jumps (ratio = 1 +- eps, pg1 == pg2)?
    with torch.no_grad():
        # epoch 0 uses the initial parameters
        pass


def gru_case(algo, E, A, T, Do, Ds, K, H, tb, tile):
    from cleanmarl_amd.gru import GRUPPOLearner
    from cleanmarl_amd.learner import DeviceBatch, HParams, NetSpec, init_params_like_torch
    if tile != "auto":
        os.environ["CM_GRU_TILE"] = tile
    torch.manual_seed(2)
    batch = t._random_case(77, E, A, T, Do, Ds, K)
    aspec, cspec = NetSpec(Do, H, 0, K, "gru"), NetSpec(Ds if algo == "mappo" else Do, 64, 1, 1)
    ap, cp = init_params_like_torch(aspec), init_params_like_torch(cspec)
    hp = dict(gamma=0.99, td_lambda=0.95, normalize_advantage=True, normalize_return=False, epochs=2, ppo_clip=0.2, entropy_coef=0.01,
              clip_gradients=0.5, optimizer="Adam", learning_rate_actor=8e-4, learning_rate_critic=8e-4, tbptt=tb)
    dev = torch.device("cuda:0")
    b = DeviceBatch.from_reference_layout(batch["obs"], batch["actions"], batch["log_probs"], batch["reward"], batch["states"], batch["avail"], batch["mask"], dev)
    L = GRUPPOLearner(algo, aspec, cspec, A, HParams(**hp), dev, actor_params=[p.clone() for p in ap], critic_params=[p.clone() for p in cp])
    recs = L.train_iteration(b, keep_grads=True)
    ret, adv, orecs = R.gru_update(ap, cp, batch, hp, algo)
    print(f"GRU {algo} E{E} A{A} T{T} Do{Do} K{K} H{H} tbptt{tb} tile={tile}")
    for e, (r, o) in enumerate(zip(recs, orecs)):
        for c, ((g, after), ost) in enumerate(zip(r["actor_steps"], o["actor_steps"])):
            report(f"epoch {e} chunk {c} actor", aspec, g.cpu().numpy(), R.flat(ost["grads"]).numpy())


def shard_case(name):
    """tests/test_hip_parity.py::test_full_size_other_configs_shard_additivity_and_oracle_shard: one epoch's actor gradient of a 16-env shard
    of config 4 against the fp32 oracle AND against the same oracle evaluated in fp64 -- who is off, the HIP pass or the fp32 CPU sums?"""
    from cleanmarl_amd import _native as N
    L, b, c = t._full_size_cfg(name)
    algo = c["algo"]
    L.compute_targets(b)
    s = N.stream_ptr()
    sb = b.shard(0, c["shard"])
    L.actor_pass(sb, s); L.critic_pass(sb, s)
    torch.cuda.synchronize()
    first = L.gbuf.clone()
    Pa = L.actor.numel()
    T = b.T
    mask = torch.arange(T)[None, :] < sb.ep_len.cpu()[:, None]
    batch = dict(obs=sb.obs.permute(0, 2, 1, 3).cpu(), actions=sb.action.permute(0, 2, 1).long().cpu(), log_probs=sb.logp.permute(0, 2, 1).cpu(),
                 reward=sb.reward.cpu(), states=sb.state.cpu(), avail=sb.avail.permute(0, 2, 1, 3).bool().cpu(), mask=mask)
    hp = dict(gamma=0.99, td_lambda=0.95, epochs=1, ppo_clip=0.2, entropy_coef=0.001, clip_gradients=-1, optimizer="Adam",
              learning_rate_actor=8e-4, learning_rate_critic=8e-4, normalize_reward=False, normalize_advantage=False, normalize_return=False)
    split = lambda flat, spec: [q.reshape(sh) for q, sh in zip(torch.split(flat.cpu(), [int(np.prod(sh)) for sh in spec.shapes()]), spec.shapes())]
    ap, cp = split(L.actor, L.actor_spec), split(L.critic_params(), L.critic_spec)
    ret, adv = sb.ret.permute(0, 2, 1).cpu(), sb.adv.permute(0, 2, 1).cpu()
    scal, ag, cg = R.mlp_epoch(ap, cp, batch, ret, adv, hp, algo)
    dbl = lambda x: x.double() if x.is_floating_point() else x
    scal64, ag64, cg64 = R.mlp_epoch([p.double() for p in ap], [p.double() for p in cp], {k: dbl(v) for k, v in batch.items()}, ret.double(), adv.double(), hp, algo)
    n = float(first[Pa + 5])
    hip = (first[:Pa] / n).cpu().numpy().astype(np.float64)
    o32, o64 = R.flat(ag).numpy().astype(np.float64), R.flat(ag64).numpy()
    mx = np.abs(o64).max()
    print(f"SHARD {name}: rows {int(mask.sum()) * sb.A}  max|g| {mx:.3e}")
    print(f"  HIP   vs fp64 oracle: {np.abs(hip - o64).max() / mx:.3e}")
    print(f"  fp32 oracle vs fp64 : {np.abs(o32 - o64).max() / mx:.3e}")
    print(f"  HIP   vs fp32 oracle: {np.abs(hip - o32).max() / mx:.3e}")
    report("HIP vs fp64 oracle", L.actor_spec, hip, o64)
    report("fp32 oracle vs fp64 oracle", L.actor_spec, o32, o64)


if __name__ == "__main__":
    which = sys.argv[1] if len(sys.argv) > 1 else "all"
    if which in ("all", "shard"):
        shard_case("cfg4")
    if which in ("all", "gru"):
        gru_case("ippo", 11, 4, 13, 37, 50, 17, 64, 5, "auto")
        gru_case("ippo", 11, 4, 13, 37, 50, 17, 64, 5, "32")
