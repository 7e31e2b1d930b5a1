"""Pin the oracle (oracle/restatement.py) against golden vectors captured from the
unmodified reference scripts (tests/golden/make_golden.py).  CPU only."""
import os

import numpy as np
import pytest
import torch

from oracle import restatement as R

MLP_CASES = [("mappo_dense", "mappo"), ("mappo_ragged_norm", "mappo"), ("mappo_deep", "mappo"), ("mappo_wide", "mappo"), ("mappo_rmsprop", "mappo"), ("ippo_sgd", "ippo"),
             ("ippo_dense", "ippo"), ("ippo_ragged_norm", "ippo")]
GRU_CASES = [("mappo_lstm_ragged", "mappo"), ("mappo_lstm_dense", "mappo"), ("ippo_lstm_ragged", "ippo")]

TOL = 2e-6  # oracle-vs-reference bar (product bar is 1e-4)


def _close(a, b, tol=TOL):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    err = np.max(np.abs(a - b) / (1.0 + np.abs(b))) if a.size else 0.0
    assert err <= tol, f"max rel-abs err {err:.3e} > {tol}"


@pytest.mark.parametrize("name,algo", MLP_CASES)
def test_mlp_update_matches_reference(golden_dir, name, algo):
    torch.set_num_threads(1)
    batch, ap, cp, hp, z = R.load_golden(os.path.join(golden_dir, name + ".npz"))
    if "b_reward_raw" in z.files:  # a2: reward normalisation
        _close(R.normalize_reward(torch.from_numpy(z["b_reward_raw"]), batch["mask"]).numpy(), z["b_reward"])
    ret, adv, recs = R.mlp_update(ap, cp, batch, hp, algo)
    _close(ret.numpy(), z["return_lambda"])
    _close(adv.numpy(), z["advantages"])
    assert len(recs) == len(z["actor_losses"]) == int(hp["epochs"])
    for e, r in enumerate(recs):
        _close(r["actor_loss"], z["actor_losses"][e])
        _close(r["critic_loss"], z["critic_losses"][e])
        _close(r["entropy"], z["entropies_bonuses"][e])
        _close(r["kl"], z["kl_divergences"][e], 5e-6)
        _close(r["clipfrac"], z["clipped_ratios"][e])
        _close(r["actor_gnorm"], z["actor_gradients"][e])
        _close(r["critic_gnorm"], z["critic_gradients"][e])
        _close(R.flat(r["actor_grads"]).numpy(), z["actor_grads"][e])
        _close(R.flat(r["critic_grads"]).numpy(), z["critic_grads"][e])
        _close(R.flat(r["actor_after"]).numpy(), z["actor_after"][e])
        _close(R.flat(r["critic_after"]).numpy(), z["critic_after"][e])


@pytest.mark.parametrize("name,algo", GRU_CASES)
def test_gru_update_matches_reference(golden_dir, name, algo):
    torch.set_num_threads(1)
    batch, ap, cp, hp, z = R.load_golden(os.path.join(golden_dir, name + ".npz"))
    ret, adv, recs = R.gru_update(ap, cp, batch, hp, algo)
    _close(ret.numpy(), z["return_lambda"])
    _close(adv.numpy(), z["advantages"])
    k = 0
    for e, r in enumerate(recs):
        _close(r["actor_loss"], z["actor_losses"][e])
        _close(r["critic_loss"], z["critic_losses"][e])
        _close(r["entropy"], z["entropies_bonuses"][e])
        _close(r["kl"], z["kl_divergences"][e], 5e-6)
        _close(r["clipfrac"], z["clipped_ratios"][e])
        _close(r["actor_gnorm"], z["actor_gradients"][e])
        _close(r["critic_gnorm"], z["critic_gradients"][e])
        for s in r["actor_steps"]:
            _close(R.flat(s["grads"]).numpy(), z["actor_grads"][k])
            _close(R.flat(s["after"]).numpy(), z["actor_after"][k])
            k += 1
        _close(R.flat(r["critic_grads"]).numpy(), z["critic_grads"][e])
        _close(R.flat(r["critic_after"]).numpy(), z["critic_after"][e])
    assert k == len(z["actor_grads"])


def test_logged_scalars_are_epoch_means(golden_dir):
    """train/* tags are means over epochs (mappo_multienvs.py:605-612)."""
    z = np.load(os.path.join(golden_dir, "mappo_dense.npz"))
    tags = list(z["log_tags"])
    vals = z["log_vals"]
    assert abs(vals[tags.index("train/actor_loss")] - z["actor_losses"].mean()) < 1e-9
    assert abs(vals[tags.index("train/critic_loss")] - z["critic_losses"].mean()) < 1e-9
    assert vals[tags.index("train/num_updates")] == 3


@pytest.mark.parametrize("name,algo", [("mappo_dense", "mappo"), ("mappo_deep", "mappo"), ("ippo_dense", "ippo")])
def test_c_td_lambda_matches_reference(golden_dir, name, algo):
    """oracle/td_lambda.c (literal per-episode reverse loop in C) against the reference's return_lambda / advantages
    (cases without advantage / return normalisation, where the golden arrays are the raw scan outputs)."""
    from oracle.build_c import td_lambda_c
    batch, ap, cp, hp, z = R.load_golden(os.path.join(golden_dir, name + ".npz"))
    with torch.no_grad():
        values = R.critic_values(cp, batch, algo).contiguous()
        values = values * batch["mask"].unsqueeze(-1)
    ret, adv = td_lambda_c(batch["reward"].numpy(), values.numpy(), batch["mask"].numpy(), hp["gamma"], hp["td_lambda"])
    _close(ret, z["return_lambda"])
    _close(adv, z["advantages"])
    r2, a2 = R.td_lambda(batch["reward"], values, batch["mask"], hp["gamma"], hp["td_lambda"])
    _close(ret, r2.numpy(), 1e-6)


@pytest.mark.parametrize("name", ["coma_tdlambda", "coma_nstep", "coma_default_width"])
def test_coma_restatement_matches_reference_golden(golden_dir, name):
    """oracle/coma.py vs the unmodified cleanmarl/coma_multienvs.py (one iteration: targets, critic step, polyak, actor step)."""
    from oracle import coma as C
    batch, ap, cp, hp, z = C.load_golden(os.path.join(golden_dir, name + ".npz"))
    if "b_reward_raw" in z.files:
        raw = torch.from_numpy(z["b_reward_raw"])
        assert np.abs(R.normalize_reward(raw, batch["mask"]).numpy() - z["b_reward"]).max() < 1e-6
    tp = [p.clone() for p in cp]
    rec = C.update(ap, cp, tp, batch, hp)
    assert np.abs(rec["ret"].numpy() - z["return_lambda"]).max() < 2e-5
    assert abs(rec["critic_loss"] - float(z["cr_loss"])) < 1e-5 * (1 + abs(float(z["cr_loss"])))
    assert abs(rec["actor_loss"] - float(z["ac_loss"])) < 1e-5 * (1 + abs(float(z["ac_loss"])))
    assert abs(rec["entropy"] - float(z["entropies"])) < 1e-6
    assert abs(rec["critic_gnorm"] - float(z["critic_gradients"])) < 1e-5 * (1 + float(z["critic_gradients"]))
    assert abs(rec["actor_gnorm"] - float(z["actor_gradients"])) < 1e-5 * (1 + float(z["actor_gradients"]))
    assert np.abs(rec["critic_grads"].numpy() - z["critic_grads"][0]).max() < 2e-6
    assert np.abs(rec["actor_grads"].numpy() - z["actor_grads"][0]).max() < 2e-6
    assert np.abs(R.flat(cp).numpy() - z["critic_after"][0]).max() < 2e-6
    assert np.abs(R.flat(ap).numpy() - z["actor_after"][0]).max() < 2e-6
    assert np.abs(R.flat(tp).numpy() - z["target_after"]).max() < 1e-7


@pytest.mark.parametrize("name,algo", MLP_CASES)
def test_plain_c_loss_loop_matches_reference_golden(golden_dir, name, algo):
    """oracle/ppo_loss.c (literal per-time-step, per-env, per-agent scalar loops of mappo_multienvs.py:527-576) reproduces the
    epoch-0 logged scalars of the unmodified reference: actor loss, critic loss, entropy, approx KL, clip fraction."""
    from oracle.build_c import ppo_losses_c
    batch, ap, cp, hp, z = R.load_golden(os.path.join(golden_dir, name + ".npz"))
    got = ppo_losses_c(batch, ap, cp, torch.from_numpy(z["advantages"]), torch.from_numpy(z["return_lambda"]), hp["ppo_clip"],
                       hp["entropy_coef"], algo)
    assert got["n_valid"] == float(batch["mask"].sum())
    for k, zk in (("actor_loss", "actor_losses"), ("critic_loss", "critic_losses"), ("entropy", "entropies_bonuses"),
                  ("kl", "kl_divergences"), ("clipfrac", "clipped_ratios")):
        assert abs(got[k] - float(z[zk][0])) <= 5e-6 * (1.0 + abs(float(z[zk][0]))), (k, got[k], float(z[zk][0]))
