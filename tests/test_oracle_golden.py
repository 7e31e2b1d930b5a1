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


OPT_CASES = [("mappo_dense", "mappo"), ("mappo_ragged_norm", "mappo"), ("mappo_deep", "mappo"), ("mappo_rmsprop", "mappo"),
             ("ippo_sgd", "ippo"), ("ippo_ragged_norm", "ippo"), ("ippo_dense", "ippo")]


@pytest.mark.parametrize("name,algo", OPT_CASES)
def test_plain_c_clip_and_optimiser_step_matches_reference_golden(golden_dir, name, algo):
    """oracle/optim_moments.c::clip_optim_step_ref (clip_grad_norm_ + one torch.optim step, all four optimisers the goldens cover:
    Adam, AdamW, SGD, RMSprop) fed with the reference's per-epoch gradients reproduces the reference's post-step parameters and its
    logged pre-clip gradient norms, epoch after epoch (optimiser state carried)."""
    from oracle.build_c import clip_optim_step_c
    batch, ap, cp, hp, z = R.load_golden(os.path.join(golden_dir, name + ".npz"))
    for net, init, lr in (("actor", ap, hp["learning_rate_actor"]), ("critic", cp, hp["learning_rate_critic"])):
        p = R.flat(init).numpy().astype(np.float32).copy()
        m, v = np.zeros_like(p), np.zeros_like(p)
        for e in range(int(hp["epochs"])):
            g = np.ascontiguousarray(z[f"{net}_grads"][e], np.float32).copy()
            # the goldens hold p.grad AFTER clip_grad_norm_ (what optimizer.step() consumed) and the logged norm BEFORE it
            norm = clip_optim_step_c(p, g, m, v, e + 1, lr, str(hp["optimizer"]), float(hp["clip_gradients"]))
            logged, mx = float(z[f"{net}_gradients"][e]), float(hp["clip_gradients"])
            _close(norm, mx * logged / (logged + 1e-6) if 0 < mx < logged else logged, 5e-6)
            _close(p, z[f"{net}_after"][e])


def test_plain_c_clip_branch_matches_torch():
    """Six goldens DO clip (mappo_ragged_norm, mappo_wide, mappo_rmsprop, ippo_ragged_norm and both COMA cases with max_norm 0.5: critic norms
    0.51 .. 3.56; test_plain_c_clip_and_optimiser_step_matches_reference_golden runs that branch against the reference's post-step
    parameters).  This test adds what they cannot: the same branch for every optimiser kind over several consecutive steps, against torch itself."""
    from oracle.build_c import clip_optim_step_c
    torch.manual_seed(0)
    for kind, cls in (("Adam", torch.optim.Adam), ("AdamW", torch.optim.AdamW), ("SGD", torch.optim.SGD), ("RMSprop", torch.optim.RMSprop)):
        ref = torch.nn.Parameter(torch.randn(1000))
        opt = cls([ref], lr=3e-3)
        p = ref.detach().numpy().copy(); m, v = np.zeros_like(p), np.zeros_like(p)
        for step in range(1, 4):
            g = torch.randn(1000) * 2.0
            ref.grad = g.clone()
            n_ref = torch.nn.utils.clip_grad_norm_([ref], 0.7)
            opt.step()
            gc = g.numpy().copy()
            n = clip_optim_step_c(p, gc, m, v, step, 3e-3, kind, 0.7)
            _close(n, float(n_ref)); _close(gc, ref.grad.numpy()); _close(p, ref.detach().numpy())


@pytest.mark.parametrize("name,algo", [("mappo_ragged_norm", "mappo"), ("ippo_ragged_norm", "ippo")])
def test_plain_c_masked_normalisation_matches_reference_golden(golden_dir, name, algo):
    """oracle/optim_moments.c::masked_normalize_ref: advantage normalisation (agent-mean moments over valid steps, unbiased std, no
    epsilon, :505-512) reproduces the reference's normalised advantages from the raw TD(lambda) ones, and the reward normalisation of
    RolloutBuffer.get_batch (:143-146, eps = 1e-6, valid entries only) the reference's b_reward."""
    from oracle.build_c import masked_normalize_c
    batch, ap, cp, hp, z = R.load_golden(os.path.join(golden_dir, name + ".npz"))
    mask = batch["mask"].numpy()
    with torch.no_grad():
        values = R.critic_values(cp, batch, algo)
        ret, adv_raw = R.td_lambda(batch["reward"], values, batch["mask"], hp["gamma"], hp["td_lambda"])
    assert hp["normalize_advantage"]
    adv_n, (cnt, mean, sd) = masked_normalize_c(adv_raw.numpy(), mask)
    assert cnt == mask.sum()
    _close(adv_n, z["advantages"], 5e-6)
    if hp["normalize_return"]:
        _close(masked_normalize_c(ret.numpy(), mask)[0], z["return_lambda"], 5e-6)
    if "b_reward_raw" in z.files:
        rn, _ = masked_normalize_c(z["b_reward_raw"][..., None], mask, eps=1e-6, valid_only=True)
        _close(rn[..., 0], z["b_reward"], 5e-6)


def test_plain_c_gru_cell_and_sampler_match_the_pinned_oracle():
    """gru_cell_ref against torch.nn.GRUCell (what cleanmarl/mappo_lstm_multienvs.py:162-184 calls) and categorical_sample_ref against
    oracle/sampling.py (the Python twin of the device sampler: same uniforms -> same actions, same log-probs)."""
    from oracle import sampling
    from oracle.build_c import categorical_sample_c, gru_cell_c
    torch.manual_seed(3)
    cell = torch.nn.GRUCell(21, 48)
    x, h = torch.randn(17, 21), torch.randn(17, 48)
    with torch.no_grad():
        want = cell(x, h).numpy()
    got = gru_cell_c(x.numpy(), h.numpy(), cell.weight_ih.detach().numpy(), cell.weight_hh.detach().numpy(), cell.bias_ih.detach().numpy(),
                     cell.bias_hh.detach().numpy())
    _close(got, want, 5e-6)
    rng = np.random.default_rng(5)
    logits = rng.normal(size=(400, 9)).astype(np.float32) * 2
    avail = rng.random((400, 9)) < 0.6
    avail[:, 2] = True
    logits = np.where(avail, logits, np.float32(-1e9))
    a_py, lp_py, u = sampling.act(logits, avail, seed=11, row_offset=100, t=7)
    a_c, lp_c = categorical_sample_c(logits, u)
    assert (a_c == a_py).all() and avail[np.arange(400), a_c].all()
    _close(lp_c, lp_py, 2e-6)
