"""The reference-structured CPU driver (bench.py's cpu_baseline) computes the same numbers as the oracle."""
import numpy as np
import torch

from oracle import reference_loop
from oracle import restatement as R


def test_reference_loop_matches_oracle():
    torch.set_num_threads(1)
    out = reference_loop.run(E=3, A=2, T=6, hidden=16, epochs=2)
    ap, cp = [p.clone() for p in out["init"][0]], [p.clone() for p in out["init"][1]]
    hp = dict(gamma=0.99, td_lambda=0.95, epochs=2, ppo_clip=0.2, entropy_coef=1e-3, clip_gradients=-1,
              optimizer="Adam", learning_rate_actor=8e-4, learning_rate_critic=8e-4)
    ret, adv, recs = R.mlp_update(ap, cp, out["batch"], hp, "mappo")
    assert np.abs(ret.numpy() - out["ret"].numpy()).max() < 1e-5
    assert np.abs(adv.numpy() - out["adv"].numpy()).max() < 1e-5
    for (al, cl), r in zip(out["logs"], recs):
        assert abs(al - r["actor_loss"]) < 1e-5 and abs(cl - r["critic_loss"]) < 1e-5
    assert np.abs(R.flat(ap).numpy() - R.flat(out["actor"]).numpy()).max() < 1e-5
    assert np.abs(R.flat(cp).numpy() - R.flat(out["critic"]).numpy()).max() < 1e-5
    assert out["agent_steps_per_s"] > 0
