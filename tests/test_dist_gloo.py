"""N > 1 path on CPU: 2 / 4 / 8 gloo processes each own a shard of the envs, build the UN-NORMALISED per-shard
gradient/statistic buffers (here with the CPU oracle standing in for the HIP kernels), run the product's
collectives (cleanmarl_amd/dist.py) and must land on the single-process full-batch result."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from cleanmarl_amd import dist
from oracle import restatement as R

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "mappo_ragged_norm.npz")


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _shard_sums(ap, cp, batch, ret, adv, hp, algo):
    """What cm_ppo_actor_fwd_bwd + cm_critic_fwd_bwd emit for one shard: gradient of the loss SUMS, + stats."""
    apr = [p.detach().clone().requires_grad_(True) for p in ap]
    cpr = [p.detach().clone().requires_grad_(True) for p in cp]
    at = R.actor_terms(R.actor_logits(apr, batch["obs"], batch["avail"]), batch, adv, hp["ppo_clip"], hp["entropy_coef"])
    vl = R.critic_term(R.critic_values(cpr, batch, algo), ret, batch["mask"])
    ag = torch.autograd.grad(at["loss"], apr)
    cg = torch.autograd.grad(vl, cpr)
    n = batch["mask"].sum().float()
    st_a = torch.stack([at["pg"], at["ent"], at["kl"], at["clip"], torch.tensor(0.0), n, torch.tensor(0.0), torch.tensor(0.0)]).detach()
    st_c = torch.stack([torch.tensor(0.0)] * 4 + [vl.detach(), n, torch.tensor(0.0), torch.tensor(0.0)])
    return torch.cat([R.flat(ag), st_a, R.flat(cg), st_c]).detach()


def _worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.distributed.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    batch, ap, cp, hp, z = R.load_golden(GOLD)
    B = batch["obs"].shape[0]
    lo, n = dist.shard(B, rank, world)
    sub = {k: v[lo:lo + n] for k, v in batch.items()}
    # targets: TD(lambda) is per-env; the normalisations need GLOBAL moments -> merge_moments_
    # a rank WITHOUT envs (world > B) contributes a zero-count triple and an all-zero buffer, but enters every collective
    with torch.no_grad():
        vals = R.critic_values(cp, sub, "mappo")
        ret, adv = R.td_lambda(sub["reward"], vals, sub["mask"], hp["gamma"], hp["td_lambda"]) if n else (torch.zeros(0), torch.zeros(0))
    outs = []
    for x in (adv, ret):
        if n:
            y = x.mean(-1)[sub["mask"]].double()
            mom = torch.stack([torch.tensor(float(y.numel()), dtype=torch.float64), y.mean(), ((y - y.mean()) ** 2).sum()])
        else:
            mom = torch.zeros(3, dtype=torch.float64)
        dist.merge_moments_(mom, None, world)
        outs.append(((x - mom[1].float()) / torch.sqrt(mom[2] / (mom[0] - 1)).float()) if n else x)
    adv, ret = outs
    Pa, Pc = R.flat(ap).numel(), R.flat(cp).numel()
    buf = _shard_sums(ap, cp, sub, ret, adv, hp, "mappo") if n else torch.zeros(Pa + Pc + 16)
    dist.allreduce_sum_(buf, None, world)
    torch.save(dict(buf=buf, adv=adv, ret=ret, lo=lo, n=n), f"{out}.{rank}")
    torch.distributed.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4, 8])  # the golden holds 7 envs: at world 8 one rank owns none
def test_env_sharding_equals_full_batch(tmp_path, world):
    port, out = _free_port(), str(tmp_path / "r")
    mp.spawn(_worker, args=(world, port, out), nprocs=world, join=True)
    parts = [torch.load(f"{out}.{r}") for r in range(world)]
    batch, ap, cp, hp, z = R.load_golden(GOLD)
    ret, adv = R.prepare_targets(batch, cp, dict(hp, normalize_advantage=True, normalize_return=True), "mappo")
    assert sum(g["n"] for g in parts) == batch["obs"].shape[0] and (world < 8 or min(g["n"] for g in parts) == 0)
    for g in parts:
        lo, n = g["lo"], g["n"]
        if n:
            assert np.abs(g["adv"].numpy() - adv[lo:lo + n].numpy()).max() < 2e-6
            assert np.abs(g["ret"].numpy() - ret[lo:lo + n].numpy()).max() < 2e-6
        assert torch.equal(g["buf"], parts[0]["buf"])  # every rank holds the same reduced buffer
    got = parts[0]
    scal, ag, cg = R.mlp_epoch(ap, cp, batch, ret, adv, hp, "mappo")
    Pa, Pc = R.flat(ap).numel(), R.flat(cp).numel()
    buf = got["buf"]
    N = buf[Pa + 5]
    assert N == batch["mask"].sum() == buf[Pa + 8 + Pc + 5]
    assert np.abs((buf[:Pa] / N).numpy() - R.flat(ag).numpy()).max() < 2e-6
    assert np.abs((buf[Pa + 8:Pa + 8 + Pc] / N).numpy() - R.flat(cg).numpy()).max() < 2e-6
    assert abs(float((-buf[Pa + 0] - hp["entropy_coef"] * buf[Pa + 1]) / N) - scal["actor_loss"]) < 2e-6
    assert abs(float(buf[Pa + 8 + Pc + 4] / N) - scal["critic_loss"]) < 2e-6


def test_shard_partition():
    for E, W in [(4096, 8), (10, 3), (2, 4), (7, 1)]:
        parts = [dist.shard(E, r, W) for r in range(W)]
        assert sum(n for _, n in parts) == E
        assert all(parts[i][0] + parts[i][1] == parts[i + 1][0] for i in range(W - 1))
