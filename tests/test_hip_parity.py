"""GPU parity tests: HIP path (through the C-ABI) vs golden vectors captured from the unmodified reference
and vs the CPU oracle on seeded inputs.  Tolerance: north_star's 1e-4 (fp32), written per assert."""
import os

import numpy as np
import pytest
import torch

from parity import DISP_TOL, GRAD_TOL, TOL, StepChecker, check_grads, check_step, disp_err, golden_before, golden_init, grad_err, twin_err  # noqa: F401
from parity import OptimizerTwin, check_grads_kink
from parity import err as _err

pytestmark = pytest.mark.gpu


MLP_CASES = [("mappo_dense", "mappo"), ("mappo_ragged_norm", "mappo"), ("mappo_deep", "mappo"), ("mappo_wide", "mappo"), ("mappo_rmsprop", "mappo"), ("ippo_sgd", "ippo"),
             ("ippo_dense", "ippo"), ("ippo_ragged_norm", "ippo")]



def _learner_from_golden(path, algo):
    from oracle import restatement as R
    from cleanmarl_amd.learner import DeviceBatch, HParams, NetSpec, PPOLearner
    batch, ap, cp, hp, z = R.load_golden(path)
    dev = torch.device("cuda:0")
    reward = torch.from_numpy(z["b_reward_raw"]) if "b_reward_raw" in z.files else batch["reward"]
    b = DeviceBatch.from_reference_layout(batch["obs"], batch["actions"], batch["log_probs"], reward,
                                          batch["states"], batch["avail"], batch["mask"], dev)
    A = batch["obs"].shape[2]
    H = HParams(gamma=hp["gamma"], td_lambda=hp["td_lambda"], normalize_reward=bool(hp["normalize_reward"]),
                normalize_advantage=bool(hp["normalize_advantage"]), normalize_return=bool(hp["normalize_return"]),
                epochs=int(hp["epochs"]), ppo_clip=hp["ppo_clip"], entropy_coef=hp["entropy_coef"],
                clip_gradients=hp["clip_gradients"], optimizer=hp["optimizer"],
                learning_rate_actor=hp["learning_rate_actor"], learning_rate_critic=hp["learning_rate_critic"])
    aspec = NetSpec(ap[0].shape[1], ap[0].shape[0], len(ap) // 2 - 2, ap[-1].shape[0])
    cspec = NetSpec(cp[0].shape[1], cp[0].shape[0], len(cp) // 2 - 2, 1)
    L = PPOLearner(algo, aspec, cspec, A, H, dev, actor_params=ap, critic_params=cp)
    return L, b, z, batch


def _check_update_against_golden(b, recs, z, label):
    """Every quantity the reference logs or produces in an update against the golden of the unmodified reference (tests/parity.py: the
    three bars).  Raises AssertionError on the first miss."""
    if "b_reward_raw" in z.files:
        assert _err(b.reward, z["b_reward"], "reward") <= TOL
    ret = b.ret.permute(0, 2, 1).cpu().numpy()
    adv = b.adv.permute(0, 2, 1).cpu().numpy()
    assert _err(ret, z["return_lambda"], "returns") <= TOL
    assert _err(adv, z["advantages"], "advantages") <= TOL
    kind = str(z["hp_optimizer"])  # the REFERENCE run's optimiser and learning rates: what the HIP steps are held to
    ca = StepChecker(golden_init(z, "actor"), kind, float(z["hp_learning_rate_actor"]), label + " actor")
    cc = StepChecker(golden_init(z, "critic"), kind, float(z["hp_learning_rate_critic"]), label + " critic")
    for e, r in enumerate(recs):
        assert _err(r["actor_loss"], z["actor_losses"][e], "losses") <= TOL
        assert _err(r["critic_loss"], z["critic_losses"][e], "losses") <= TOL
        assert _err(r["entropy"], z["entropies_bonuses"][e], "statistics") <= TOL
        assert _err(r["kl"], z["kl_divergences"][e], "statistics") <= TOL
        assert _err(r["clipfrac"], z["clipped_ratios"][e], "statistics") <= TOL
        assert grad_err(r["actor_gnorm"], z["actor_gradients"][e], "gnorm") <= GRAD_TOL
        assert grad_err(r["critic_gnorm"], z["critic_gradients"][e], "gnorm") <= GRAD_TOL
        ca.step(r["actor_grads"], r["actor_after"], z["actor_grads"][e], z["actor_after"][e])
        cc.step(r["critic_grads"], r["critic_after"], z["critic_grads"][e], z["critic_after"][e])


@pytest.mark.parametrize("name,algo", MLP_CASES)
def test_update_matches_reference_golden(golden_dir, name, algo):
    L, b, z, batch = _learner_from_golden(os.path.join(golden_dir, name + ".npz"), algo)
    recs = L.train_iteration(b, keep_grads=True)
    _check_update_against_golden(b, recs, z, "mlp golden")


@pytest.mark.parametrize("what,factor", [("learning_rate_actor", 1.05), ("learning_rate_critic", 0.95), ("entropy_coef", 10.0), ("td_lambda", 0.99), ("gamma", 0.99)])
def test_a_wrong_learner_fails_the_golden_comparison(golden_dir, what, factor):
    """The bars can fail: a learner whose optimiser step is 5 % too long (or short), whose entropy bonus is ten times too large or whose
    lambda / gamma is off by 1 % does NOT pass the comparison that test_update_matches_reference_golden applies (under the old absolute 1e-4 on the parameters a
    12 %-wrong Adam step passed: max|after - before| is 8e-4 in this golden)."""
    L, b, z, batch = _learner_from_golden(os.path.join(golden_dir, "mappo_dense.npz"), "mappo")
    setattr(L.hp, what, getattr(L.hp, what) * factor)
    L.opt_a.lr, L.opt_c.lr = L.hp.learning_rate_actor, L.hp.learning_rate_critic
    recs = L.train_iteration(b, keep_grads=True)
    with pytest.raises(AssertionError):
        _check_update_against_golden(b, recs, z, "deliberately wrong")


def _random_case(seed, E, A, T, Do, Ds, K, ragged=True, avail_p=0.7):
    g = torch.Generator().manual_seed(seed)
    obs = torch.randn(E, T, A, Do, generator=g)
    states = torch.randn(E, T, Ds, generator=g)
    avail = torch.rand(E, T, A, K, generator=g) < avail_p
    avail[..., 0] = True
    probs = avail.float() / avail.float().sum(-1, keepdim=True)
    actions = torch.multinomial(probs.reshape(-1, K), 1, generator=g).reshape(E, T, A)
    logp = -torch.rand(E, T, A, generator=g) * 2.0
    reward = torch.randn(E, T, generator=g)
    if ragged:
        lens = torch.randint(max(1, T // 2), T + 1, (E,), generator=g)
        lens[0] = T
    else:
        lens = torch.full((E,), T)
    mask = torch.arange(T)[None, :] < lens[:, None]
    m = mask[..., None]
    obs = obs * m[..., None]; states = states * m; avail = avail & m[..., None]
    actions = actions * mask[..., None]; logp = logp * m; reward = reward * mask
    return dict(obs=obs, actions=actions, log_probs=logp, reward=reward, states=states, avail=avail, mask=mask)


@pytest.mark.parametrize("algo,E,A,T,Do,Ds,K,H,L", [
    ("mappo", 37, 3, 25, 21, 54, 5, 64, 1),     # config-1/2 shapes, ragged tile edge
    ("mappo", 16, 8, 32, 56, 384, 5, 64, 1),    # config-3 shapes (6 input chunks for the critic)
    # the register-resident wide-input critic (csrc/cm_critic_fused.h) at every chunk count it serves, ragged rows, narrow H
    ("mappo", 11, 3, 13, 21, 150, 5, 64, 1),    # 3 chunks (config-5 state width), 4-float aligned after padding only
    ("mappo", 7, 2, 29, 24, 200, 4, 48, 1),     # 4 chunks, H = 48
    ("mappo", 5, 3, 41, 24, 300, 4, 64, 1),     # 5 chunks
    ("ippo", 6, 3, 20, 140, 10, 5, 64, 1),      # per-agent critic on a 140-wide observation (3 chunks), 3-chunk actor
    ("mappo", 4, 11, 23, 24, 260, 4, 64, 1),    # 11 agents: targets beyond the 8 prefetched ones
    ("mappo", 300, 2, 30, 24, 448, 4, 64, 1),   # 7 chunks, 282 row tiles over 256 workgroups (second pass of the tile loop)
    ("ippo", 12, 10, 40, 115, 243, 17, 64, 1),  # config-4 shapes (2 chunks, 17 actions, avail masks)
    ("mappo", 9, 2, 17, 70, 140, 6, 32, 2),     # H=32 padded to 64, 2 hidden layers, 2/3 chunks
    ("ippo", 5, 3, 8, 7, 11, 3, 48, 0),         # no hidden->hidden layer, odd H
    # the layered schedule (csrc/cm_mlp_wide.h): wider than 64 units or deeper than 2 hidden->hidden layers
    ("mappo", 21, 3, 19, 21, 54, 5, 128, 1),    # the reference's COMA-critic default width
    ("ippo", 12, 4, 15, 37, 50, 17, 96, 3),     # 1.5 slabs of 64 units, 3 hidden layers, odd input width (scalar loads), 17 actions
    ("mappo", 9, 2, 17, 70, 140, 6, 200, 0),    # no hidden->hidden layer, 200 units
    ("mappo", 7, 3, 11, 21, 54, 5, 32, 4),      # narrow but deep
    ("mappo", 5, 2, 300, 24, 48, 5, 256, 2),    # widest supported, several row tiles
    # more than 32 actions (SMAC's 27m_vs_30m has 36): the layered schedule's 64-wide head, whatever the hidden width
    ("ippo", 12, 4, 15, 37, 50, 36, 64, 1),     # a shape the fused kernels would serve but for its head
    ("mappo", 9, 3, 17, 70, 140, 64, 128, 1),   # widest head; 128 units with one hidden->hidden layer stay OFF the fused 128-wide tile
    ("ippo", 6, 3, 20, 140, 10, 33, 96, 2),     # one action past the fused head, 1.5 slabs, wide observations
])
def test_update_matches_oracle_seeded(algo, E, A, T, Do, Ds, K, H, L):
    _seeded_case(algo, E, A, T, Do, Ds, K, H, L, normalize=True)


@pytest.fixture(autouse=True)
def _fused_critic_at_test_sizes(monkeypatch):
    """The one-pass wide-input critic (csrc/cm_critic_fused.h) is selected from 131072 rows on; the cases of this module are far
    smaller, so they force it wherever its shape rules allow (129..448 aligned input columns, one hidden layer) -- the two-kernel
    schedule it replaces keeps its own cases below."""
    monkeypatch.setenv("CM_CRITIC_SCHEDULE", "fused")
    from cleanmarl_amd import _native as N
    N.sync_env_options()  # the C library never reads the environment: the package maps its CM_* hooks onto cm_set_option


@pytest.mark.parametrize("algo,E,A,T,Do,Ds,K,H,L", [("mappo", 16, 8, 32, 56, 384, 5, 64, 1), ("mappo", 7, 2, 29, 24, 200, 4, 48, 1),
                                                      ("ippo", 6, 3, 20, 140, 10, 5, 64, 1)])
def test_wide_critic_split_schedule_matches_oracle(algo, E, A, T, Do, Ds, K, H, L, monkeypatch):
    monkeypatch.setenv("CM_CRITIC_SCHEDULE", "split")  # what batches below 131072 rows run
    _seeded_case(algo, E, A, T, Do, Ds, K, H, L, normalize=True)


@pytest.mark.gpu
@pytest.mark.parametrize("E,A,T,Ds,H", [(16, 8, 32, 384, 64), (7, 2, 29, 200, 48), (40, 3, 130, 475, 64)])
def test_streaming_dw0_batch_sizes_leave_the_same_bits(E, A, T, Ds, H, monkeypatch):
    """k_dw0_stream with 16 rows in flight per lane (the default above 2^16 rows) and with 8 (what fits beside the six-wave rollout, the
    default below) accumulate the same rows in the same order: bit-identical critic gradients and post-step parameters on the split
    schedule; the last case has four 32-column tiles per wave (the 16-row form spills there)."""
    from cleanmarl_amd import _native as N
    from cleanmarl_amd.learner import DeviceBatch, HParams, NetSpec, PPOLearner, init_params_like_torch
    monkeypatch.setenv("CM_CRITIC_SCHEDULE", "split")
    dev = torch.device("cuda:0")
    batch = _random_case(321, E, A, T, 24, Ds, 5)
    aspec, cspec = NetSpec(24, 64, 1, 5), NetSpec(Ds, H, 1, 1)
    torch.manual_seed(3)
    ap, cp = init_params_like_torch(aspec), init_params_like_torch(cspec)
    outs = []
    for bsz in ("8", "4"):
        monkeypatch.setenv("CM_DW0_BATCH", bsz)
        N.sync_env_options()
        assert N.load().cm_get_option(b"dw0_batch") == bsz.encode()
        b = DeviceBatch.from_reference_layout(batch["obs"], batch["actions"], batch["log_probs"], batch["reward"], batch["states"],
                                              batch["avail"], batch["mask"], dev)
        Lr = PPOLearner("mappo", aspec, cspec, A, HParams(epochs=2), dev, actor_params=[p.clone() for p in ap],
                        critic_params=[p.clone() for p in cp])
        recs = Lr.train_iteration(b, keep_grads=True)
        outs.append([(r["critic_grads"].clone(), r["critic_after"].clone()) for r in recs])
    for (g8, p8), (g4, p4) in zip(*outs):
        assert torch.equal(g8, g4) and torch.equal(p8, p4)
        assert g8.abs().sum().item() > 0


@pytest.mark.gpu
@pytest.mark.parametrize("algo,E,A,T,Do,Ds,K,H,L", [
    ("mappo", 21, 3, 19, 21, 54, 5, 128, 1),    # actor + critic inside one launch each (inputs <= 64 columns, odd widths: scalar tile loads)
    ("mappo", 16, 8, 32, 56, 384, 5, 128, 1),   # config-3 shapes: critic on the 384-wide state (layer 0 outside: k_wide_gemm + k_dw0_stream)
    ("ippo", 12, 4, 15, 37, 50, 17, 96, 1),     # 96 units (zero-padded slab), 17 actions (two head column blocks), per-agent critic
    ("ippo", 6, 3, 20, 140, 10, 32, 65, 1),     # 65 units, 140-wide observations (actor AND critic with layer 0 outside), 32 actions
    ("mappo", 3, 2, 131, 64, 64, 8, 128, 1),    # 64-column inputs exactly, several row tiles with a ragged tail
])
def test_fused_128_wide_tile_matches_oracle_and_the_layered_schedule(algo, E, A, T, Do, Ds, K, H, L, monkeypatch):
    """csrc/cm_mlp_fused128.h (65..128 hidden units, one hidden->hidden layer: the reference's COMA critic default and any
    --*_hidden_dim up to 128) against the oracle at the module's bar, and against the layered schedule it replaces (option
    wide_schedule = layered) -- two different summation orders of the same fp32 products."""
    from cleanmarl_amd import _native as N
    monkeypatch.setenv("CM_WIDE_SCHEDULE", "fused")
    N.sync_env_options()
    assert N.load().cm_get_option(b"wide_schedule") == b"fused"
    ef = _seeded_case(algo, E, A, T, Do, Ds, K, H, L, normalize=True)
    monkeypatch.setenv("CM_WIDE_SCHEDULE", "layered")
    N.sync_env_options()
    assert N.load().cm_get_option(b"wide_schedule") == b"layered"
    el = _seeded_case(algo, E, A, T, Do, Ds, K, H, L, normalize=True)
    assert ef != el or all(v == 0.0 for v in ef.values())  # the option really switched kernels (bit-different sums)


def _seeded_case(algo, E, A, T, Do, Ds, K, H, L, normalize, tol=TOL, pad=False):
    from oracle import restatement as R
    from cleanmarl_amd.learner import DeviceBatch, HParams, NetSpec, PPOLearner, init_params_like_torch
    torch.manual_seed(1)
    batch = _random_case(123, E, A, T, Do, Ds, K)
    aspec = NetSpec(Do, H, L, K)
    cspec = NetSpec(Ds if algo == "mappo" else Do, H, L, 1)
    ap = init_params_like_torch(aspec)
    cp = init_params_like_torch(cspec)
    hp = dict(gamma=0.99, td_lambda=0.95, normalize_advantage=normalize, normalize_return=False, epochs=2, ppo_clip=0.2,
              entropy_coef=0.01, clip_gradients=0.5, optimizer="Adam", learning_rate_actor=8e-4, learning_rate_critic=8e-4)
    dev = torch.device("cuda:0")
    b = DeviceBatch.from_reference_layout(batch["obs"], batch["actions"], batch["log_probs"], batch["reward"],
                                          batch["states"], batch["avail"], batch["mask"], dev, pad=pad)
    Lr = PPOLearner(algo, aspec, cspec, A, HParams(**hp), dev, actor_params=[p.clone() for p in ap],
                    critic_params=[p.clone() for p in cp])
    recs = Lr.train_iteration(b, keep_grads=True)
    before = {"actor": R.flat(ap).clone(), "critic": R.flat(cp).clone()}  # (mlp_update steps the lists in place)
    ret, adv, orecs = R.mlp_update(ap, cp, batch, hp, algo)
    errs = {"ret": _err(b.ret.permute(0, 2, 1).cpu().numpy(), ret.numpy()), "adv": _err(b.adv.permute(0, 2, 1).cpu().numpy(), adv.numpy())}
    scale = tol / TOL  # callers with a looser tier (bf16) loosen every bar by the same factor
    bars = {}
    twins = {"actor": OptimizerTwin(before["actor"], hp["optimizer"], hp["learning_rate_actor"]),
             "critic": OptimizerTwin(before["critic"], hp["optimizer"], hp["learning_rate_critic"])}
    for epoch, (r, o) in enumerate(zip(recs, orecs)):
        for k in ("actor_loss", "critic_loss", "entropy", "kl", "clipfrac"):
            errs[k] = max(errs.get(k, 0.0), _err(r[k], o[k], "seeded " + k))
        for k in ("actor_gnorm", "critic_gnorm"):
            errs[k] = max(errs.get(k, 0.0), grad_err(r[k], o[k], "seeded gnorm"))
            bars[k] = GRAD_TOL * scale
        for net in ("actor", "critic"):
            k = net + ("_grads" if epoch == 0 else "_grads_later")  # later epochs: gradients at parameters that drifted apart (StepChecker)
            errs[k] = max(errs.get(k, 0.0), grad_err(r[net + "_grads"], R.flat(o[net + "_grads"]), "seeded " + k))
            bars[k] = GRAD_TOL * scale * (1.0 if epoch == 0 else StepChecker.LATER_GRAD)
            k = net + "_twin"  # the HIP step against torch.optim's on the gradient it consumed: every entry, whatever the arithmetic of the passes
            errs[k] = max(errs.get(k, 0.0), twin_err(twins[net], r[net + "_grads"], r[net + "_after"], "seeded " + k))
            bars[k] = DISP_TOL
            k = net + "_after"  # ... and against the oracle's parameters: the absolute end-to-end bound (tests/parity.py: why not the displacement)
            errs[k] = max(errs.get(k, 0.0), _err(r[k], R.flat(o[k]), "seeded " + k))
            bars[k] = tol
            disp_err(r[k], R.flat(o[k]), before[net], "seeded " + k + " displacement (well-conditioned entries, recorded)", ref_grad=R.flat(o[net + "_grads"]))
            before[net] = R.flat(o[k])
    for k, v in errs.items():
        assert v <= bars.get(k, tol), (k, v)
    return errs


BF16X3_GRAD = 1e-3


def test_bf16x3_opt_in_keeps_the_parity_bar():
    """CM_MFMA=bf16x3 (error-compensated bf16 MFMA loops, docs/KERNEL_NOTES.md section 8; the env var is read once per process, hence the
    subprocess): the config-3-shaped update must stay inside the same 1e-4 bar vs the fp32 oracle, and must really have
    taken the bf16 kernels (cm_mfma_mode() == 1; results differ from the exact-fp32 run in the low bits)."""
    import json
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    code = ("import sys, json; sys.path.insert(0, %r); sys.path.insert(0, %r); import test_hip_parity as t; "
            "from cleanmarl_amd import _native as N; "
            "e = t._seeded_case('mappo', 48, 8, 64, 56, 384, 5, 64, 1, True, tol=1e9); e['mode'] = N.load().cm_mfma_mode(); print('ERRS ' + json.dumps(e))"
            % (here, os.path.dirname(here)))
    out = {}
    for mode in ("bf16x3", ""):
        env = dict(os.environ, CM_MFMA=mode)
        p = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
        assert p.returncode == 0, p.stderr[-2000:]
        out[mode] = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("ERRS ")][0][5:])
    assert out["bf16x3"]["mode"] == 1 and out[""]["mode"] == 0
    # the compensated mode's tier: returns / advantages / losses / statistics / post-step parameters inside north_star's 1e-4; gradients
    # within 1e-3 of the largest gradient entry (the exact-fp32 default: 1e-4); the optimiser step itself at the twin's bar in every mode
    for k, v in out[""].items():  # the exact-fp32 default on the same case: the three bars of tests/parity.py
        if k != "mode":
            assert v <= (GRAD_TOL * (StepChecker.LATER_GRAD if k.endswith("_later") else 1.0) if "grad" in k or "gnorm" in k else DISP_TOL if k.endswith("_twin") else TOL), (k, v)
    for k, v in out["bf16x3"].items():
        if k != "mode":  # (the optimiser step itself is the same fp32 kernel in every mode: the twin bar does not loosen)
            assert v <= (BF16X3_GRAD if "grad" in k or "gnorm" in k else DISP_TOL if k.endswith("_twin") else TOL), (k, v)
    assert out["bf16x3"]["actor_grads"] != out[""]["actor_grads"]
    print("max errors vs oracle  fp32:", {k: f"{v:.1e}" for k, v in out[""].items() if k != "mode"})
    print("max errors vs oracle bf16x3:", {k: f"{v:.1e}" for k, v in out["bf16x3"].items() if k != "mode"})


def test_bf16_single_pass_has_its_own_parity_tier():
    """CM_MFMA=bf16 (option mfma=bf16: ONE bf16 MFMA per product, operands rounded to 8 mantissa bits, fp32 accumulate -- SURVEY 8(f)-4's
    fast path, opt-in): NOT inside north_star's 1e-4 bar by construction, so it gets its own tier, asserted here on the config-3-shaped
    update: returns / advantages (fp32 value pass) unchanged at 1e-4, losses / entropy / KL / clip fraction within 2e-2, gradient
    norms and gradients within 5e-2, post-Adam parameters within 1e-2; the learning-signal test runs with it as well (test_cli_gpu)."""
    import json
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    code = ("import sys, json; sys.path.insert(0, %r); sys.path.insert(0, %r); import test_hip_parity as t; "
            "from cleanmarl_amd import _native as N; "
            "e = t._seeded_case('mappo', 48, 8, 64, 56, 384, 5, 64, 1, True, tol=1e9); e['mode'] = N.load().cm_mfma_mode(); print('ERRS ' + json.dumps(e))"
            % (here, os.path.dirname(here)))
    p = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, CM_MFMA="bf16"), capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    e = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("ERRS ")][0][5:])
    print("max errors vs oracle, single-pass bf16:", {k: f"{v:.1e}" for k, v in e.items() if k != "mode"})
    assert e["mode"] == 2
    assert e["ret"] <= TOL and e["adv"] <= TOL
    for k in ("actor_loss", "critic_loss", "entropy", "kl", "clipfrac"):
        assert e[k] <= 2e-2, (k, e[k])
    for k in ("actor_gnorm", "critic_gnorm", "actor_grads", "critic_grads", "actor_grads_later", "critic_grads_later"):
        assert e[k] <= 5e-2, (k, e[k])
    for k in ("actor_after", "critic_after"):
        assert e[k] <= 1e-2, (k, e[k])
    for k in ("actor_twin", "critic_twin"):    # the optimiser step is the fp32 kernel of every mode
        assert e[k] <= DISP_TOL, (k, e[k])
    assert e["actor_grads"] > 1e-6  # it really took the single-pass kernels


def test_scan_long_sequences_and_edge_lengths():
    """T not a multiple of 64, T > 64 lanes, ep_len in {0, 1, T}."""
    from oracle import restatement as R
    from cleanmarl_amd import _native as N
    lib = N.load()
    dev = torch.device("cuda:0")
    for (E, A, Av, T) in [(7, 3, 1, 1), (5, 4, 4, 63), (9, 2, 1, 150), (6, 3, 3, 257)]:
        g = torch.Generator().manual_seed(E * 1000 + T)
        reward = torch.randn(E, T, generator=g)
        values = torch.randn(E, Av, T, generator=g)
        lens = torch.randint(0, T + 1, (E,), generator=g)
        lens[0] = T
        if E > 1:
            lens[1] = 0
        if E > 2:
            lens[2] = 1
        ret = torch.empty(E, A, T, device=dev); adv = torch.empty(E, A, T, device=dev)
        d_r, d_v, d_l = reward.to(dev), values.to(dev), lens.int().to(dev)  # keep alive: raw pointers cross the ABI
        N.check(lib.cm_td_lambda_scan(N.ptr(d_r), N.ptr(d_v), N.ptr(d_l), E, A, Av, T,
                                      0.99, 0.95, N.ptr(ret), N.ptr(adv), N.stream_ptr()), "scan")
        mask = torch.arange(T)[None, :] < lens[:, None]
        v_ref = values.permute(0, 2, 1).expand(E, T, A)
        r_ref, a_ref = R.td_lambda(reward * mask, v_ref, mask, 0.99, 0.95)
        assert _err(ret.permute(0, 2, 1).cpu().numpy(), r_ref.numpy()) <= TOL
        assert _err(adv.permute(0, 2, 1).cpu().numpy(), a_ref.numpy()) <= TOL


def test_mlp_forward_matches_oracle():
    from oracle import restatement as R
    from cleanmarl_amd import _native as N
    from cleanmarl_amd.learner import NetSpec, flatten_params, init_params_like_torch
    lib = N.load()
    dev = torch.device("cuda:0")
    torch.manual_seed(3)
    for (rows, din, H, L, dout) in [(1, 5, 64, 1, 1), (1000, 384, 64, 1, 1), (777, 115, 64, 1, 17), (130, 21, 32, 2, 5), (64, 200, 17, 0, 32)]:
        spec = NetSpec(din, H, L, dout)
        p = init_params_like_torch(spec)
        x = torch.randn(rows, din)
        avail = torch.rand(rows, dout) < 0.6
        y = torch.empty(rows, dout, device=dev)
        d_x, d_p, d_a = x.to(dev), flatten_params(p, dev), avail.to(torch.uint8).to(dev)
        N.check(lib.cm_mlp_forward(N.ptr(d_x), rows, din, H, L, dout, N.ptr(d_p), N.ptr(d_a), N.ptr(y), N.stream_ptr()), "fwd")
        ref = R.actor_logits(p, x, avail)
        assert _err(y.cpu().numpy(), ref.numpy()) <= TOL
    # layered schedule through the workspace entry point (which forwards narrow shapes to the fused kernel with ws = NULL)
    for (rows, din, H, L, dout) in [(1000, 384, 128, 1, 1), (130, 21, 256, 2, 5), (77, 115, 100, 3, 17), (300, 56, 64, 1, 5), (1, 9, 65, 0, 32)]:
        spec = NetSpec(din, H, L, dout)
        p = init_params_like_torch(spec)
        x = torch.randn(rows, din)
        avail = torch.rand(rows, dout) < 0.6
        y = torch.empty(rows, dout, device=dev)
        d_x, d_p, d_a = x.to(dev), flatten_params(p, dev), avail.to(torch.uint8).to(dev)
        need = lib.cm_mlp_forward_workspace_bytes(rows, din, H, L, dout)
        assert (need == 0) == (H <= 64 and L <= 2)
        ws = torch.empty(max(need, 1), dtype=torch.uint8, device=dev)
        N.check(lib.cm_mlp_forward_ws(N.ptr(d_x), rows, din, H, L, dout, N.ptr(d_p), N.ptr(d_a), N.ptr(y), N.ptr(ws) if need else None,
                                      need, N.stream_ptr()), "fwd_ws")
        assert _err(y.cpu().numpy(), R.actor_logits(p, x, avail).numpy()) <= TOL
        if need:
            assert lib.cm_mlp_forward_ws(N.ptr(d_x), rows, din, H, L, dout, N.ptr(d_p), N.ptr(d_a), N.ptr(y), N.ptr(ws), need - 1,
                                         N.stream_ptr()) != 0 and b"workspace too small" in lib.cm_last_error()


def test_fused_128_forward_transposed_kernel_matches_oracle_and_the_round3_forward(monkeypatch):
    """k_mlp128_fwd (csrc/cm_mlp_fused128.h, round 4: transposed products, H1 in registers, partial logits summed in wave order, two
    workgroups per CU) against the oracle and against the forward of the training tile it replaced (option wide_schedule = fused_r3):
    inputs inside (<= 64 columns) and outside (k_wide_gemm -> z0) the launch, 65 / 96 / 128 units, heads of 1 .. 32 outputs (one and two
    column blocks, partial-slab strides 4 .. 32), ragged last tiles, more tiles than resident workgroups."""
    from oracle import restatement as R
    from cleanmarl_amd import _native as N
    from cleanmarl_amd.learner import NetSpec, flatten_params, init_params_like_torch
    lib = N.load()
    dev = torch.device("cuda:0")
    torch.manual_seed(5)
    differ = 0
    for (rows, din, H, dout) in [(1000, 384, 128, 1), (777, 56, 128, 5), (513, 33, 96, 17), (70, 64, 65, 32), (40000, 56, 128, 5), (129, 7, 128, 16)]:
        spec = NetSpec(din, H, 1, dout)
        p = init_params_like_torch(spec)
        x = torch.randn(rows, din)
        avail = torch.rand(rows, dout) < 0.7
        d_x, d_p, d_a = x.to(dev), flatten_params(p, dev), avail.to(torch.uint8).to(dev)
        need = lib.cm_mlp_forward_workspace_bytes(rows, din, H, 1, dout)
        ws = torch.empty(max(need, 1), dtype=torch.uint8, device=dev)
        ys = []
        for sched in ("fused", "fused_r3"):
            monkeypatch.setenv("CM_WIDE_SCHEDULE", sched)
            N.sync_env_options()
            assert lib.cm_get_option(b"wide_schedule") == sched.encode()
            y = torch.full((rows, dout), float("nan"), device=dev)
            N.check(lib.cm_mlp_forward_ws(N.ptr(d_x), rows, din, H, 1, dout, N.ptr(d_p), N.ptr(d_a), N.ptr(y), N.ptr(ws), need, N.stream_ptr()), "fwd_ws")
            ys.append(y.cpu())
        ref = R.actor_logits(p, x, avail)
        for y in ys:
            assert _err(y.numpy(), ref.numpy()) <= TOL
        differ += int(not torch.equal(ys[0], ys[1]))
    assert differ > 0  # the option really switched kernels (different summation orders)
    monkeypatch.delenv("CM_WIDE_SCHEDULE")
    N.sync_env_options()


def test_unknown_optimizer_fails_loudly():
    from cleanmarl_amd import _native as N
    from cleanmarl_amd.learner import HParams, NetSpec, PPOLearner
    with pytest.raises(N.NativeError, match="Adagrad"):
        PPOLearner("mappo", NetSpec(8, 32, 1, 3), NetSpec(16, 32, 1, 1), 2, HParams(optimizer="Adagrad"), torch.device("cuda:0"))


def test_unsupported_shapes_fail_loudly():
    from cleanmarl_amd import _native as N
    lib = N.load()
    rc = lib.cm_mlp_forward(None, 10, 8, 4096, 1, 1, None, None, None, N.stream_ptr())
    assert rc != 0 and b"hidden_dim" in lib.cm_last_error()


def test_env_kernels_match_numpy_twin():
    """cm_synth_env_reset / cm_synth_env_step vs cleanmarl_amd/env/synthetic.py on identical seeds + actions."""
    from cleanmarl_amd import _native as N
    from cleanmarl_amd.env.synthetic import SyntheticSpreadEnv
    lib = N.load()
    dev = torch.device("cuda:0")
    for (E, A, T, ids, off) in [(5, 3, 6, 1, 0), (70, 8, 4, 1, 1000), (3, 2, 5, 0, 7)]:
        Do, Ds = 6 * A + ids * A, 6 * A * A
        g = torch.Generator().manual_seed(E)
        action = torch.randint(0, 5, (E, A, T), generator=g, dtype=torch.int32).to(dev)
        es = torch.zeros(E, 6 * A, device=dev)
        obs = torch.zeros(E, A, T, Do, device=dev); state = torch.zeros(E, T, Ds, device=dev); rew = torch.zeros(E, T, device=dev)
        s = N.stream_ptr()
        N.check(lib.cm_synth_env_reset(N.ptr(es), E, A, ids, 42, off, 3, N.ptr(obs), N.ptr(state), T, s), "reset")
        for t in range(T):
            N.check(lib.cm_synth_env_step(N.ptr(es), N.ptr(action), E, A, ids, t, T, N.ptr(rew), N.ptr(obs), N.ptr(state), s), "step")
        obs_c, state_c, rew_c, act_c = obs.cpu().numpy(), state.cpu().numpy(), rew.cpu().numpy(), action.cpu().numpy()
        for e in range(E):
            env = SyntheticSpreadEnv(A, bool(ids), max_cycles=T, seed=42, env_index=off + e)
            env.episode = 2  # next reset() is episode 3
            o, _ = env.reset()
            for t in range(T):
                assert np.abs(obs_c[e, :, t] - o).max() <= 1e-5
                assert np.abs(state_c[e, t] - env.get_state()).max() <= 1e-5
                o, r, done, trunc, _ = env.step(act_c[e, :, t])
                assert abs(rew_c[e, t] - r) <= 1e-4
            assert trunc and not done


def test_policy_act_matches_cpu_sampler():
    """cm_policy_act (strided, in-place rollout-buffer addressing) vs oracle/sampling.py."""
    import ctypes as C
    from oracle import restatement as R
    from oracle import sampling
    from cleanmarl_amd import _native as N
    from cleanmarl_amd.learner import NetSpec, flatten_params, init_params_like_torch
    lib = N.load()
    dev = torch.device("cuda:0")
    torch.manual_seed(5)
    E, A, T, Do, K, t = 33, 5, 7, 40, 9, 3
    spec = NetSpec(Do, 64, 1, K)
    p = init_params_like_torch(spec)
    obs = torch.randn(E, A, T, Do)
    avail = torch.rand(E, A, T, K) < 0.5
    avail[..., 2] = True
    d_obs, d_av, d_p = obs.to(dev), avail.to(torch.uint8).to(dev), flatten_params(p, dev)
    action = torch.full((E, A, T), -7, dtype=torch.int32, device=dev)
    logp = torch.full((E, A, T), 9.0, device=dev)
    N.check(lib.cm_policy_act(C.c_void_p(d_obs.data_ptr() + 4 * t * Do), T * Do, C.c_void_p(d_av.data_ptr() + t * K), T * K,
                              E * A, Do, 64, 1, K, N.ptr(d_p), 99, 1234, t,
                              C.c_void_p(action.data_ptr() + 4 * t), C.c_void_p(logp.data_ptr() + 4 * t), T, N.stream_ptr()), "act")
    logits = R.actor_logits(p, obs[:, :, t].reshape(E * A, Do), avail[:, :, t].reshape(E * A, K)).numpy()
    a_ref, lp_ref, u = sampling.act(logits, avail[:, :, t].reshape(E * A, K).numpy(), 99, 1234, t)
    a_gpu = action[:, :, t].reshape(-1).cpu().numpy()
    lp_gpu = logp[:, :, t].reshape(-1).cpu().numpy()
    same = a_gpu == a_ref
    assert same.mean() >= 0.99  # a uniform within fp32 round-off of a CDF edge may flip a neighbour
    assert np.abs(lp_gpu[same] - lp_ref[same]).max() <= TOL
    assert avail[:, :, t].reshape(E * A, K).numpy()[np.arange(E * A), a_gpu].all()  # never an unavailable action
    untouched = torch.ones(T, dtype=torch.bool); untouched[t] = False
    assert (action[:, :, untouched] == -7).all() and (logp[:, :, untouched] == 9.0).all()


@pytest.mark.parametrize("H,L,K", [(64, 1, 9), (128, 1, 9), (200, 3, 9), (32, 4, 9), (64, 1, 36), (128, 1, 64)])
def test_policy_act_ws_matches_cpu_sampler_for_fused_and_layered_shapes(H, L, K):
    """cm_policy_act_ws (the three act modes behind one entry point; layered schedule for wide / deep actors) vs oracle/sampling.py."""
    import ctypes as C
    from oracle import restatement as R
    from oracle import sampling
    from cleanmarl_amd import _native as N
    from cleanmarl_amd.learner import NetSpec, flatten_params, init_params_like_torch
    lib = N.load()
    dev = torch.device("cuda:0")
    torch.manual_seed(6)
    E, A, T, Do, t = 29, 5, 7, 40, 4
    spec = NetSpec(Do, H, L, K)
    p = init_params_like_torch(spec)
    obs = torch.randn(E, A, T, Do)
    avail = torch.rand(E, A, T, K) < 0.5
    avail[..., 2] = True
    d_obs, d_av, d_p = obs.to(dev), avail.to(torch.uint8).to(dev), flatten_params(p, dev)
    need = lib.cm_policy_act_workspace_bytes(E * A, Do, H, L, K)
    assert (need == 0) == (H <= 64 and L <= 2 and K <= 32)  # heads wider than 32 actions: layered schedule (64-wide logits plane)
    ws = torch.empty(max(need, 1), dtype=torch.uint8, device=dev)
    av_t = avail[:, :, t].reshape(E * A, K).numpy()
    logits = R.actor_logits(p, obs[:, :, t].reshape(E * A, Do), avail[:, :, t].reshape(E * A, K)).numpy()
    for eps in (0.0, 0.3, -1.0):
        action = torch.full((E, A, T), -7, dtype=torch.int32, device=dev)
        logp = torch.full((E, A, T), 9.0, device=dev)
        N.check(lib.cm_policy_act_ws(C.c_void_p(d_obs.data_ptr() + 4 * t * Do), T * Do, C.c_void_p(d_av.data_ptr() + t * K), T * K,
                                     E * A, Do, H, L, K, N.ptr(d_p), eps, 99, 1234, t, C.c_void_p(action.data_ptr() + 4 * t),
                                     C.c_void_p(logp.data_ptr() + 4 * t), T, N.ptr(ws) if need else None, need, N.stream_ptr()), "act_ws")
        a_gpu = action[:, :, t].reshape(-1).cpu().numpy()
        lp_gpu = logp[:, :, t].reshape(-1).cpu().numpy()
        if eps < 0:
            a_ref = logits.argmax(-1)
            lp_ref = (logits - np.log(np.exp(logits - logits.max(-1, keepdims=True)).sum(-1, keepdims=True)) - logits.max(-1, keepdims=True))[
                np.arange(E * A), a_ref]
        elif eps > 0:
            a_ref, lp_ref, _ = sampling.act_eps(logits, av_t, eps, 99, 1234, t)
        else:
            a_ref, lp_ref, _ = sampling.act(logits, av_t, 99, 1234, t)
        same = a_gpu == a_ref
        assert same.mean() >= 0.99
        assert np.abs(lp_gpu[same] - lp_ref[same]).max() <= TOL
        assert av_t[np.arange(E * A), a_gpu].all()
        untouched = torch.ones(T, dtype=torch.bool); untouched[t] = False
        assert (action[:, :, untouched] == -7).all() and (logp[:, :, untouched] == 9.0).all()


@pytest.mark.parametrize("env_type", ["synthetic", "synthetic_cpu"])
def test_wide_actor_rollout_and_training_run_end_to_end(env_type, tmp_path, monkeypatch):
    """--actor_hidden_dim=128 --critic_hidden_dim=128 --critic_num_layers=3 through the CLI driver: per-step rollout with the
    layered act, layered value pass and updates; finite logs and a changed policy."""
    import math
    from cleanmarl_amd.driver import run
    monkeypatch.chdir(tmp_path)
    out = run("mappo_multienvs", [f"--env_type={env_type}", "--batch_size=8", "--synthetic_steps=10", "--total_timesteps=400",
                                  "--actor_hidden_dim=128", "--critic_hidden_dim=128", "--critic_num_layers=3", "--log_every=1",
                                  "--eval_steps=2", "--num_eval_ep=2"])
    assert all(math.isfinite(v) for _, v, _ in out["history"])
    assert out["training_step"] >= 3 and "eval/ep_reward" in {t for t, _, _ in out["history"]}


@pytest.mark.parametrize("script,env_type", [("ippo_multienvs", "synthetic_shape"), ("mappo_multienvs", "synthetic_shape_cpu"),
                                             ("mappo_lstm_multienvs", "synthetic_shape")])
def test_scripts_run_with_more_than_32_actions(script, env_type, tmp_path, monkeypatch):
    """36 actions (SMAC's 27m_vs_30m: 6 + 30 enemies) through the CLI at the reference's default network widths: the act passes, the
    actor's training pass and greedy evaluation leave the fused kernels (heads up to 32 actions) for the layered schedule's 64-wide head."""
    import math
    from cleanmarl_amd.driver import run
    monkeypatch.chdir(tmp_path)
    out = run(script, [f"--env_type={env_type}", "--batch_size=6", "--synthetic_agents=4", "--synthetic_steps=9", "--synthetic_obs=40",
                       "--synthetic_state=60", "--synthetic_actions=36", "--total_timesteps=300", "--log_every=1", "--eval_steps=2",
                       "--num_eval_ep=2"])
    assert all(math.isfinite(v) for _, v, _ in out["history"])
    assert out["training_step"] >= 3 and "eval/ep_reward" in {t for t, _, _ in out["history"]}


@pytest.mark.parametrize("tile", [64, "64s", 16, "16s"])
@pytest.mark.parametrize("E,A,T,H,L", [(37, 8, 12, 64, 1), (50, 3, 9, 64, 1), (10, 5, 6, 32, 0), (21, 9, 7, 64, 1)])
def test_fused_rollout_matches_per_step_kernels(E, A, T, H, L, tile, monkeypatch):
    """cm_rollout_spread (one persistent launch; every tiling: 64-row workgroup tiles, the 16-row form and its store-wave variant that
    small env counts take) == reset + T x (cm_policy_act, cm_synth_env_step)."""
    from cleanmarl_amd.learner import NetSpec, flatten_params, init_params_like_torch
    from cleanmarl_amd.rollout import SyntheticSpreadRollout
    monkeypatch.setenv("CM_ROLLOUT_TILE", str(tile))
    dev = torch.device("cuda:0")
    torch.manual_seed(11)
    ra = SyntheticSpreadRollout(E, A, T, seed=7, device=dev, env_offset=123)
    rb = SyntheticSpreadRollout(E, A, T, seed=7, device=dev, env_offset=123)
    spec = NetSpec(ra.Do, H, L, 5)
    p = flatten_params(init_params_like_torch(spec), dev)
    for _ in range(2):  # two consecutive episodes (episode counter + action keys advance identically)
        ba = ra.collect(p, spec, fused=True)
        bb = rb.collect(p, spec, fused=False)
    torch.cuda.synchronize()
    same_act = (ba.action == bb.action)
    # identical arithmetic up to FMA contraction: a flipped sample can only happen at a CDF edge and then the
    # env trajectory of that one env diverges; require step 0 exact and >= 99% of all (env,agent,t) equal
    assert torch.equal(ba.obs[:, :, 0], bb.obs[:, :, 0]) and torch.equal(ba.state[:, 0], bb.state[:, 0])
    assert same_act[:, :, 0].all()
    assert same_act.float().mean().item() >= 0.99
    env_ok = same_act.all(dim=2).all(dim=1)  # envs whose whole action sequence agrees
    assert env_ok.float().mean().item() >= 0.9
    assert (ba.obs[env_ok] - bb.obs[env_ok]).abs().max().item() <= 1e-5
    assert (ba.state[env_ok] - bb.state[env_ok]).abs().max().item() <= 1e-5
    assert (ba.reward[env_ok] - bb.reward[env_ok]).abs().max().item() <= 1e-4
    assert (ba.logp[env_ok] - bb.logp[env_ok]).abs().max().item() <= 1e-4
    assert (ra.env_state[env_ok] - rb.env_state[env_ok]).abs().max().item() <= 1e-5


@pytest.mark.parametrize("eps", [0.0, 0.3])
@pytest.mark.parametrize("E,A,T,H,L", [(37, 8, 12, 64, 1), (50, 3, 9, 64, 1), (10, 5, 6, 32, 0), (21, 9, 7, 64, 1), (3, 1, 5, 48, 1), (512, 8, 128, 64, 1)])
def test_rollout_tilings_are_bit_identical(E, A, T, H, L, eps, monkeypatch):
    """The tiling -- hence the env count of a GPU's shard -- never changes a trajectory: 64-row and 16-row tiles, each as a four-wave
    workgroup and as the six-wave form (writer + scorer waves), feed the MFMA the same k order, sample with the same serial sums and draw the same Philox words, so every
    buffer of two consecutive episodes is bit-identical (T not a multiple of 4 exercises the store wave's partial action / log-prob
    flush, 9 agents the scalar state stores (54 floats per segment), eps > 0 COMA's mixture sampler)."""
    from cleanmarl_amd.learner import NetSpec, flatten_params, init_params_like_torch
    from cleanmarl_amd.rollout import SyntheticSpreadRollout
    dev = torch.device("cuda:0")
    torch.manual_seed(3)
    outs = {}
    for tile in ("64", "64s", "16", "16s"):
        monkeypatch.setenv("CM_ROLLOUT_TILE", tile)
        r = SyntheticSpreadRollout(E, A, T, seed=7, device=dev, env_offset=5)
        spec = NetSpec(r.Do, H, L, 5)
        torch.manual_seed(3)
        p = flatten_params(init_params_like_torch(spec), dev)
        got = []
        for _ in range(2):
            b = r.collect(p, spec, fused=True, eps=eps)
            torch.cuda.synchronize()
            got.append({k: getattr(b, k).clone() for k in ("obs", "state", "action", "logp", "reward")})
        got.append({"env_state": r.env_state.clone()})
        outs[tile] = got
    for tile in ("64s", "16", "16s"):
        for ep, (x, y) in enumerate(zip(outs["64"], outs[tile])):
            for k in x:
                assert torch.equal(x[k], y[k]), (tile, ep, k)


@pytest.mark.parametrize("E,A,T,H", [(37, 5, 12, 64), (50, 3, 9, 32), (9, 8, 7, 64), (3, 1, 5, 48), (300, 5, 11, 64)])
def test_fused_gru_rollout_six_wave_equals_four_wave(E, A, T, H, monkeypatch):
    """The six-wave fused GRU rollout (writer + scorer waves off the step chain, the default) leaves the SAME BITS in every buffer as
    the four-wave kernel (rollout_tile = 16); the last shape has more tiles than compute units (the tile loop runs twice)."""
    from cleanmarl_amd.gru import GRUSyntheticRollout
    from cleanmarl_amd.learner import NetSpec, flatten_params, init_params_like_torch
    dev = torch.device("cuda:0")
    torch.manual_seed(14)
    spec = NetSpec(6 * A + A, H, 0, 5, "gru")
    p = flatten_params(init_params_like_torch(spec), dev)
    outs = {}
    for tile in ("16", "auto"):
        if tile == "auto":
            monkeypatch.delenv("CM_ROLLOUT_TILE", raising=False)
        else:
            monkeypatch.setenv("CM_ROLLOUT_TILE", tile)
        r = GRUSyntheticRollout(E, A, T, seed=5, device=dev, env_offset=77)
        eps = []
        for _ in range(2):
            b = r.collect(p, spec, fused=True)
            torch.cuda.synchronize()
            eps.append({k: getattr(b, k).clone() for k in ("obs", "state", "action", "logp", "reward")} | {"env": r.env_state.clone()})
        outs[tile] = eps
    for ep, (x, y) in enumerate(zip(outs["16"], outs["auto"])):
        for k in x:
            assert torch.equal(x[k], y[k]), (ep, k)


@pytest.mark.parametrize("tile", ["auto", "16"])
@pytest.mark.parametrize("E,A,T,H", [(37, 5, 12, 64), (50, 3, 9, 32), (9, 8, 7, 64), (3, 1, 5, 48)])
def test_fused_gru_rollout_matches_per_step_kernels(E, A, T, H, tile, monkeypatch):
    """cm_gru_rollout_spread (one persistent launch, 32-row tiles, weights register-resident; six-wave default and the four-wave
    kernel) == reset + T x (cm_gru_policy_act, cm_synth_env_step) with the same seeds."""
    from cleanmarl_amd.gru import GRUSyntheticRollout
    from cleanmarl_amd.learner import NetSpec, flatten_params, init_params_like_torch
    if tile == "auto":
        monkeypatch.delenv("CM_ROLLOUT_TILE", raising=False)
    else:
        monkeypatch.setenv("CM_ROLLOUT_TILE", tile)
    dev = torch.device("cuda:0")
    torch.manual_seed(13)
    ra = GRUSyntheticRollout(E, A, T, seed=5, device=dev, env_offset=77)
    rb = GRUSyntheticRollout(E, A, T, seed=5, device=dev, env_offset=77)
    spec = NetSpec(ra.Do, H, 0, 5, "gru")
    p = flatten_params(init_params_like_torch(spec), dev)
    for _ in range(2):
        ba = ra.collect(p, spec, fused=True)
        bb = rb.collect(p, spec, fused=False)
    torch.cuda.synchronize()
    same_act = (ba.action == bb.action)
    assert torch.equal(ba.obs[:, :, 0], bb.obs[:, :, 0]) and torch.equal(ba.state[:, 0], bb.state[:, 0])
    assert same_act[:, :, 0].all()
    assert same_act.float().mean().item() >= 0.99
    env_ok = same_act.all(dim=2).all(dim=1)
    assert env_ok.float().mean().item() >= 0.9
    assert (ba.obs[env_ok] - bb.obs[env_ok]).abs().max().item() <= 1e-5
    assert (ba.state[env_ok] - bb.state[env_ok]).abs().max().item() <= 1e-5
    assert (ba.reward[env_ok] - bb.reward[env_ok]).abs().max().item() <= 1e-4
    assert (ba.logp[env_ok] - bb.logp[env_ok]).abs().max().item() <= 1e-4
    assert (ra.env_state[env_ok] - rb.env_state[env_ok]).abs().max().item() <= 1e-5


GRU_CASES = [("mappo_lstm_ragged", "mappo"), ("mappo_lstm_dense", "mappo"), ("ippo_lstm_ragged", "ippo")]


@pytest.mark.parametrize("tile", ["auto", "split", "nosplit", "32", "64"])
@pytest.mark.parametrize("name,algo", GRU_CASES)
def test_gru_tbptt_update_matches_reference_golden(golden_dir, name, algo, tile, monkeypatch):
    # "auto" takes the pipelined 32-row sweeps at this batch size with the forward sweep split at its dependence on h inside the workgroup,
    # "split" the same split as a pre-pass launch (k_gru2_pre), "nosplit" the unsplit sweeps of round 5; CM_GRU_TILE=32 the four-wave kernels; CM_GRU_TILE=64 forces the 64-row streaming kernels that large batches use: all three are pinned to the goldens
    if tile != "auto":
        monkeypatch.setenv("CM_GRU_TILE", tile)
    else:
        monkeypatch.delenv("CM_GRU_TILE", raising=False)
    """cm_gru_actor_chunk_fwd_bwd + TBPTT schedule vs the unmodified reference's mappo/ippo_lstm_multienvs.py."""
    from oracle import restatement as R
    from cleanmarl_amd.gru import GRUPPOLearner
    from cleanmarl_amd.learner import DeviceBatch, HParams, NetSpec
    batch, ap, cp, hp, z = R.load_golden(os.path.join(golden_dir, name + ".npz"))
    dev = torch.device("cuda:0")
    b = DeviceBatch.from_reference_layout(batch["obs"], batch["actions"], batch["log_probs"], batch["reward"],
                                          batch["states"], batch["avail"], batch["mask"], dev)
    A = batch["obs"].shape[2]
    H = HParams(gamma=hp["gamma"], td_lambda=hp["td_lambda"], normalize_reward=bool(hp["normalize_reward"]),
                normalize_advantage=bool(hp["normalize_advantage"]), normalize_return=bool(hp["normalize_return"]),
                epochs=int(hp["epochs"]), ppo_clip=hp["ppo_clip"], entropy_coef=hp["entropy_coef"],
                clip_gradients=hp["clip_gradients"], optimizer=hp["optimizer"], tbptt=int(hp["tbptt"]),
                learning_rate_actor=hp["learning_rate_actor"], learning_rate_critic=hp["learning_rate_critic"])
    aspec = NetSpec(ap[0].shape[1], ap[0].shape[0], 0, ap[-1].shape[0], "gru")
    cspec = NetSpec(cp[0].shape[1], cp[0].shape[0], len(cp) // 2 - 2, 1)
    L = GRUPPOLearner(algo, aspec, cspec, A, H, dev, actor_params=ap, critic_params=cp)
    recs = L.train_iteration(b, keep_grads=True)
    assert _err(b.ret.permute(0, 2, 1).cpu().numpy(), z["return_lambda"]) <= TOL
    assert _err(b.adv.permute(0, 2, 1).cpu().numpy(), z["advantages"]) <= TOL
    k = 0
    kind = str(z["hp_optimizer"])
    ca = StepChecker(golden_init(z, "actor"), kind, float(z["hp_learning_rate_actor"]), "gru golden actor")
    cc = StepChecker(golden_init(z, "critic"), kind, float(z["hp_learning_rate_critic"]), "gru golden critic")
    for e, r in enumerate(recs):
        assert _err(r["actor_loss"], z["actor_losses"][e]) <= TOL
        assert _err(r["critic_loss"], z["critic_losses"][e]) <= TOL
        assert _err(r["entropy"], z["entropies_bonuses"][e]) <= TOL
        assert _err(r["kl"], z["kl_divergences"][e]) <= TOL
        assert _err(r["clipfrac"], z["clipped_ratios"][e]) <= TOL
        assert grad_err(r["actor_gnorm"], z["actor_gradients"][e]) <= GRAD_TOL
        assert grad_err(r["critic_gnorm"], z["critic_gradients"][e]) <= GRAD_TOL
        for g, after in r["actor_steps"]:
            ca.step(g, after, z["actor_grads"][k], z["actor_after"][k])
            k += 1
        cc.step(r["critic_grads"], r["critic_after"], z["critic_grads"][e], z["critic_after"][e])
    assert k == len(z["actor_grads"])


def _check_gru_against_oracle(recs, orecs, a0, c0, hp, label):
    """Per-epoch scalars, per-chunk actor gradients / steps and the critic's steps of a GRU update against oracle.restatement.gru_update
    (a0 / c0: flat copies of the parameters both started from, taken BEFORE the oracle stepped its lists in place)."""
    from oracle import restatement as R
    ca = StepChecker(a0, hp["optimizer"], hp["learning_rate_actor"], label + " actor")
    cc = StepChecker(c0, hp["optimizer"], hp["learning_rate_critic"], label + " critic")
    for r, o in zip(recs, orecs):
        for k in ("actor_loss", "critic_loss", "entropy", "kl", "clipfrac"):
            assert _err(r[k], o[k], label + " " + k) <= TOL, k
        for k in ("actor_gnorm", "critic_gnorm"):
            assert grad_err(r[k], o[k], label + " gnorm") <= GRAD_TOL, k
        assert len(r["actor_steps"]) == len(o["actor_steps"])
        for (g, after), ost in zip(r["actor_steps"], o["actor_steps"]):
            ca.step(g, after, R.flat(ost["grads"]), R.flat(ost["after"]))
        cc.step(r["critic_grads"], r["critic_after"], R.flat(o["critic_grads"]), R.flat(o["critic_after"]))


@pytest.mark.parametrize("tile", ["auto", "32", "64"])
@pytest.mark.parametrize("algo,E,A,T,Do,Ds,K,H,tb", [("ippo", 11, 4, 13, 37, 50, 17, 64, 5), ("mappo", 9, 3, 10, 21, 54, 5, 48, 4),
                                                      ("mappo", 40, 5, 12, 35, 150, 5, 64, 10)])
def test_gru_update_matches_oracle_seeded(algo, E, A, T, Do, Ds, K, H, tb, tile, monkeypatch):
    """Seeded GRU / TBPTT update (ragged episodes, availability masks, K = 17 heads, H < 64, several tiles) vs the CPU oracle,
    on both tilings (32-row sweeps; CM_GRU_TILE=64 -> 64-row streaming kernels)."""
    from oracle import restatement as R
    from cleanmarl_amd.gru import GRUPPOLearner
    from cleanmarl_amd.learner import DeviceBatch, HParams, NetSpec, init_params_like_torch
    if tile != "auto":
        monkeypatch.setenv("CM_GRU_TILE", tile)
    else:
        monkeypatch.delenv("CM_GRU_TILE", raising=False)
    torch.manual_seed(2)
    batch = _random_case(77, E, A, T, Do, Ds, K)
    aspec = NetSpec(Do, H, 0, K, "gru")
    cspec = NetSpec(Ds if algo == "mappo" else Do, 64, 1, 1)
    ap, cp = init_params_like_torch(aspec), init_params_like_torch(cspec)
    hp = dict(gamma=0.99, td_lambda=0.95, normalize_advantage=True, normalize_return=False, epochs=2, ppo_clip=0.2, entropy_coef=0.01,
              clip_gradients=0.5, optimizer="Adam", learning_rate_actor=8e-4, learning_rate_critic=8e-4, tbptt=tb)
    dev = torch.device("cuda:0")
    b = DeviceBatch.from_reference_layout(batch["obs"], batch["actions"], batch["log_probs"], batch["reward"], batch["states"],
                                          batch["avail"], batch["mask"], dev)
    L = GRUPPOLearner(algo, aspec, cspec, A, HParams(**hp), dev, actor_params=[p.clone() for p in ap], critic_params=[p.clone() for p in cp])
    recs = L.train_iteration(b, keep_grads=True)
    a0, c0 = R.flat(ap).clone(), R.flat(cp).clone()
    ret, adv, orecs = R.gru_update(ap, cp, batch, hp, algo)
    assert _err(b.ret.permute(0, 2, 1).cpu().numpy(), ret.numpy()) <= TOL and _err(b.adv.permute(0, 2, 1).cpu().numpy(), adv.numpy()) <= TOL
    _check_gru_against_oracle(recs, orecs, a0, c0, hp, "gru seeded")


@pytest.mark.parametrize("E,A,T,Do,K,H,t0,t1", [(40, 5, 23, 35, 5, 64, 0, 10), (40, 5, 23, 35, 5, 64, 20, 23), (11, 4, 13, 37, 17, 64, 5, 10),
                                                  (9, 3, 10, 21, 5, 48, 3, 4), (70, 3, 12, 18, 5, 64, 0, 7), (33, 2, 9, 64, 30, 40, 1, 9)])
def test_gru_forward_sweeps_agree(E, A, T, Do, K, H, t0, t1):
    """The three schedules of the 32-row tiling on one chunk: the eight-wave forward kernel (gru_tile = 8w: workspace stores and the
    head on helper waves, behind the chain) does the arithmetic of the four-wave one (gru_tile = 32) in the same order -- gradient,
    statistics and h_out must be EQUAL; the pipelined sweeps ("auto" at these sizes: the head of the forward sweep and five of the
    seven weight-gradient products of the backward sweep on other CUs, fed through agent-scope stores and step flags) sum the same
    terms per helper workgroup instead of per tile -- the gradient agrees to rounding, h_out is EQUAL, and five launches in a row give
    the same bits.  Chunks of even and odd length (a half-filled head pass), one step (nothing published), ragged episodes, several
    tiles, K > 16 (the 32-wide head)."""
    from cleanmarl_amd import _native as N
    from cleanmarl_amd.learner import NetSpec, flatten_params, init_params_like_torch
    lib, dev = N.load(), torch.device("cuda:0")
    torch.manual_seed(4)
    batch = _random_case(5, E, A, T, Do, 8, K)
    spec = NetSpec(Do, H, 0, K, "gru")
    params = flatten_params(init_params_like_torch(spec), dev)
    obs = batch["obs"].permute(0, 2, 1, 3).contiguous().float().to(dev)       # [E,A,T,Do]
    avail = batch["avail"].permute(0, 2, 1, 3).contiguous().to(torch.uint8).to(dev)
    act = batch["actions"].permute(0, 2, 1).contiguous().to(torch.int32).to(dev)
    lpo = batch["log_probs"].permute(0, 2, 1).contiguous().float().to(dev)
    adv = torch.randn(E, A, T, device=dev)
    eplen = batch["mask"].sum(1).to(torch.int32).to(dev)
    h0 = torch.randn(E * A, H, device=dev) * 0.3
    P = params.numel()
    wsb = lib.cm_gru_workspace_bytes(E, A, Do, H, K, t1 - t0)

    def run(tile):
        N.set_option("gru_tile", tile)
        g = torch.zeros(P + N.NUM_STATS, device=dev)
        h1 = torch.zeros(E * A, H, device=dev)
        ws = torch.zeros(wsb // 4 + 16, device=dev)
        N.check(lib.cm_gru_actor_chunk_fwd_bwd(N.ptr(obs), N.ptr(avail), N.ptr(act), N.ptr(lpo), N.ptr(adv), N.ptr(eplen), E, A, T, t0, t1, Do,
                                               H, K, N.ptr(params), N.ptr(h0), N.ptr(h1), 0.2, 0.01, N.ptr(g), N.ptr(ws), wsb, N.stream_ptr()),
                "gru chunk")
        torch.cuda.synchronize()
        return g, h1
    try:
        g4, h4 = run("32")
        g8, h8 = run("8w")
        gx, hx = run("auto")
        gn, hn_ = run("split")
        gu, hu = run("nosplit")
        again = [run("auto") for _ in range(5)]
    finally:
        N.set_option("gru_tile", "auto")
    assert torch.isfinite(g4).all()
    assert torch.equal(g8, g4) and torch.equal(h8, h4)
    assert torch.equal(hx, h4)
    scale = g4.abs().max().item()
    assert (gx - g4).abs().max().item() <= 1e-5 * (1.0 + scale)
    for g, h1 in again:
        assert torch.equal(g, gx) and torch.equal(h1, hx)
    # the forward sweep split at its dependence on h -- inside the workgroup ("auto": helper waves run fc1 and the W_ih products a step ahead of
    # the recurrence waves) or as a throughput launch in front of the chain ("split": k_gru2_pre) -- runs the MFMA sequences of the unsplit
    # pipelined sweep ("nosplit") on the same operands: the same bits everywhere
    assert torch.equal(gn, gx) and torch.equal(hn_, hx)
    assert torch.equal(gu, gx) and torch.equal(hu, hx)


def test_gru_policy_act_matches_oracle():
    from oracle import restatement as R
    from oracle import sampling
    from cleanmarl_amd import _native as N
    from cleanmarl_amd.learner import NetSpec, flatten_params, init_params_like_torch
    lib = N.load()
    dev = torch.device("cuda:0")
    torch.manual_seed(9)
    rows, Do, Hd, K = 150, 35, 64, 5
    spec = NetSpec(Do, Hd, 0, K, "gru")
    p = init_params_like_torch(spec)
    x = torch.randn(rows, Do); h0 = torch.randn(rows, Hd) * 0.5
    avail = torch.rand(rows, K) < 0.6
    avail[:, 1] = True
    d_x, d_h, d_av, d_p = x.to(dev), h0.clone().to(dev), avail.to(torch.uint8).to(dev), flatten_params(p, dev)
    act = torch.empty(rows, dtype=torch.int32, device=dev); lp = torch.empty(rows, device=dev)
    N.check(lib.cm_gru_policy_act(N.ptr(d_x), Do, N.ptr(d_av), K, rows, Do, Hd, K, N.ptr(d_p), N.ptr(d_h), 5, 77, 3,
                                  N.ptr(act), N.ptr(lp), 1, N.stream_ptr()), "gru act")
    logits, h1 = R.gru_actor_logits(p, x, h0, avail)
    assert _err(d_h.cpu().numpy(), h1.numpy()) <= TOL
    a_ref, lp_ref, _ = sampling.act(logits.numpy(), avail.numpy(), 5, 77, 3)
    same = act.cpu().numpy() == a_ref
    assert same.mean() >= 0.99
    assert np.abs(lp.cpu().numpy()[same] - lp_ref[same]).max() <= TOL


# ----------------------------------------------------------------------------------------------------------------
# Full-size (BASELINE.json configs[2]: 4096 envs x 8 agents x 128 steps) size-independent properties
# ----------------------------------------------------------------------------------------------------------------
def _full_size_setup(E=4096, A=8, T=128, K=5):
    from cleanmarl_amd.learner import DeviceBatch, HParams, NetSpec, PPOLearner, init_params_like_torch
    dev = torch.device("cuda:0")
    Do, Ds = 7 * A, 6 * A * A
    g = torch.Generator(device="cpu").manual_seed(0)
    b = DeviceBatch(E, A, T, Do, Ds, K, dev)
    b.obs.copy_(torch.randn(E, A, T, Do, generator=g)); b.state.copy_(torch.randn(E, T, Ds, generator=g))
    b.avail.fill_(1); b.action.copy_(torch.randint(0, K, (E, A, T), generator=g).int())
    b.logp.copy_(-1.6 + 0.1 * torch.randn(E, A, T, generator=g)); b.reward.copy_(torch.randn(E, T, generator=g))
    b.ep_len.copy_(torch.randint(T // 2, T + 1, (E,), generator=g).int())
    torch.manual_seed(1)
    aspec, cspec = NetSpec(Do, 64, 1, K), NetSpec(Ds, 64, 1, 1)
    L = PPOLearner("mappo", aspec, cspec, A, HParams(), dev, init_params_like_torch(aspec), init_params_like_torch(cspec))
    return L, b


def _shard_view(b, lo, hi):
    return b.shard(lo, hi)


def test_full_size_env_sharding_additivity():
    """Gradient / statistic SUMS of env shards add up to the full batch (the property multi-GPU relies on), checked at the headline
    size where no CPU oracle finishes in seconds -- and ONE of the shards (32 envs of the 4096) is small enough for the oracle:
    its sums match the CPU restatement, which pins the full-size pass to the oracle through the additivity."""
    from oracle import restatement as R
    from cleanmarl_amd import _native as N
    L, b = _full_size_setup()
    L.compute_targets(b)
    s = N.stream_ptr()
    L.actor_pass(b, s); L.critic_pass(b, s)
    full = L.gbuf.clone()
    parts = torch.zeros_like(full)
    first = None
    for lo, hi in ((0, 32), (32, 1500), (1500, 4096)):  # uneven shards, tile-unaligned boundaries
        sb = _shard_view(b, lo, hi)
        L.actor_pass(sb, s); L.critic_pass(sb, s)
        parts += L.gbuf
        if first is None:
            first = L.gbuf.clone()
    torch.cuda.synchronize()
    scale = full.abs().max().item()
    assert (full - parts).abs().max().item() <= 1e-4 * scale
    Pa, Pc = L.actor.numel(), L.critic.numel()
    assert full[Pa + 5].item() == b.ep_len.sum().item()  # N = b_mask.sum()
    # ---- the 32-env shard against the oracle (same returns / advantages as inputs: the value pass + scan have their own tests)
    sb = _shard_view(b, 0, 32)
    T = b.T
    mask = torch.arange(T)[None, :] < sb.ep_len.cpu()[:, None]
    batch = dict(obs=sb.obs.permute(0, 2, 1, 3).cpu(), actions=sb.action.permute(0, 2, 1).long().cpu(), log_probs=sb.logp.permute(0, 2, 1).cpu(),
                 reward=sb.reward.cpu(), states=sb.state.cpu(), avail=sb.avail.permute(0, 2, 1, 3).bool().cpu(), mask=mask)
    hp = dict(gamma=0.99, td_lambda=0.95, epochs=1, ppo_clip=0.2, entropy_coef=0.001, clip_gradients=-1, optimizer="Adam",
              learning_rate_actor=8e-4, learning_rate_critic=8e-4)
    ap = [p.cpu() for p in torch.split(L.actor.cpu(), [int(np.prod(sh)) for sh in L.actor_spec.shapes()])]
    ap = [p.reshape(sh) for p, sh in zip(ap, L.actor_spec.shapes())]
    cp = [p.reshape(sh) for p, sh in zip(torch.split(L.critic_params().cpu(), [int(np.prod(sh)) for sh in L.critic_spec.shapes()]), L.critic_spec.shapes())]
    scal, ag, cg = R.mlp_epoch(ap, cp, batch, sb.ret.permute(0, 2, 1).cpu(), sb.adv.permute(0, 2, 1).cpu(), hp, "mappo")
    n = float(first[Pa + 5])
    assert n == float(mask.sum())
    # gradients against the nearer of the fp32 and the fp64 evaluation of the restatement (tests/parity.py: KINK_GRAD_TOL)
    dbl = lambda v: v.double() if torch.is_tensor(v) and v.is_floating_point() else v
    _, ag64, cg64 = R.mlp_epoch([p.double() for p in ap], [p.double() for p in cp], {k: dbl(v) for k, v in batch.items()},
                                sb.ret.permute(0, 2, 1).cpu().double(), sb.adv.permute(0, 2, 1).cpu().double(), hp, "mappo")
    check_grads_kink(first[:Pa] / n, R.flat(ag), R.flat(ag64), "full-size shard actor grad")
    check_grads_kink(first[Pa + 8:Pa + 8 + Pc] / n, R.flat(cg), R.flat(cg64), "full-size shard critic grad")
    assert abs(float((-first[Pa + 0] - hp["entropy_coef"] * first[Pa + 1]) / n) - scal["actor_loss"]) <= TOL * (1 + abs(scal["actor_loss"]))
    assert abs(float(first[Pa + 8 + Pc + 4] / n) - scal["critic_loss"]) <= TOL * (1 + abs(scal["critic_loss"]))


def test_full_size_three_epochs_against_the_oracle():
    """BASELINE configs[2] at FULL size (4096 envs x 8 agents x 128 steps, ragged episodes), the whole iteration -- value pass, scan,
    THREE epochs with their optimiser steps -- against the oracle on the host cores (a batched torch-CPU restatement: about a minute on
    the GPU box): returns / advantages at 1e-4; per epoch the losses / statistics at 1e-4, the gradients relative to their largest entry,
    and every optimiser step against torch.optim on the gradient it consumed (tests/parity.py).  The oracle is TEACHER-FORCED: epoch e is
    evaluated at the parameters the HIP path holds before epoch e -- at this size an Adam step moves the ill-conditioned entries (gradient
    sums of 524 288 rows that cancel to 1e-7 of the largest entry) by a good part of lr either way, and an oracle stepped on its own drifts
    away in exactly those entries (measured: second-epoch critic gradient 6e-4 apart).  The other full-size tests pin ONE epoch's sums
    through shard additivity; this one covers what three epochs leave behind."""
    from oracle import restatement as R
    L, b = _full_size_setup()
    split = lambda flat, spec: [q.reshape(sh).clone() for q, sh in zip(torch.split(flat.detach().cpu(), [int(np.prod(sh)) for sh in spec.shapes()]), spec.shapes())]
    ap, cp = split(L.actor, L.actor_spec), split(L.critic_params(), L.critic_spec)
    recs = L.train_iteration(b, keep_grads=True)
    torch.cuda.synchronize()
    T = b.T
    mask = torch.arange(T)[None, :] < b.ep_len.cpu()[:, None]
    batch = dict(obs=b.obs.permute(0, 2, 1, 3).cpu(), actions=b.action.permute(0, 2, 1).long().cpu(), log_probs=b.logp.permute(0, 2, 1).cpu(),
                 reward=b.reward.cpu(), states=b.state.cpu(), avail=b.avail.permute(0, 2, 1, 3).bool().cpu(), mask=mask)
    hp = dict(gamma=L.hp.gamma, td_lambda=L.hp.td_lambda, epochs=L.hp.epochs, ppo_clip=L.hp.ppo_clip, entropy_coef=L.hp.entropy_coef,
              clip_gradients=L.hp.clip_gradients, optimizer=L.hp.optimizer, learning_rate_actor=L.hp.learning_rate_actor,
              learning_rate_critic=L.hp.learning_rate_critic, normalize_reward=False, normalize_advantage=False, normalize_return=False)
    assert hp["epochs"] == 3 and len(recs) == 3
    torch.set_num_threads(max(1, min(64, os.cpu_count() or 1)))
    ret, adv = R.prepare_targets(batch, cp, hp, "mappo")   # fp32, as the reference computes them
    # the per-epoch sums also in FP64 (the restatement is dtype-agnostic): at this size a gradient is held to the nearer of the two
    # evaluations of the reference's expression, tests/parity.py: KINK_GRAD_TOL
    dbl = lambda v: v.double() if torch.is_tensor(v) and v.is_floating_point() else v
    batch64 = {k: dbl(v) for k, v in batch.items()}
    ret64, adv64 = b.ret.permute(0, 2, 1).cpu().double(), b.adv.permute(0, 2, 1).cpu().double()  # (held to the fp32 oracle's just below)
    m3 = mask[:, :, None].numpy()
    assert _err(b.ret.permute(0, 2, 1).cpu().numpy() * m3, ret.numpy() * m3, "full-size returns") <= TOL
    assert _err(b.adv.permute(0, 2, 1).cpu().numpy() * m3, adv.numpy() * m3, "full-size advantages") <= TOL
    twin_a = OptimizerTwin(R.flat(ap), hp["optimizer"], hp["learning_rate_actor"])
    twin_c = OptimizerTwin(R.flat(cp), hp["optimizer"], hp["learning_rate_critic"])
    for e, r in enumerate(recs):
        scal, ag, cg = R.mlp_epoch(ap, cp, batch, ret, adv, hp, "mappo")  # the oracle at the HIP path's parameters before this epoch: as the reference computes ...
        _, ag64, cg64 = R.mlp_epoch([p.double() for p in ap], [p.double() for p in cp], batch64, ret64, adv64, hp, "mappo")  # ... and the expression's value
        for k in ("actor_loss", "critic_loss", "entropy", "kl", "clipfrac"):
            assert _err(r[k], scal[k], "full-size " + k) <= TOL, (e, k)
        assert grad_err(r["actor_gnorm"], R.grad_norm(ag).item(), "full-size gnorm") <= GRAD_TOL and grad_err(r["critic_gnorm"], R.grad_norm(cg).item(), "full-size gnorm") <= GRAD_TOL, e
        check_grads_kink(r["actor_grads"], R.flat(ag), R.flat(ag64), "full-size 3-epoch actor grad (teacher-forced)")
        check_grads_kink(r["critic_grads"], R.flat(cg), R.flat(cg64), "full-size 3-epoch critic grad (teacher-forced)")
        assert twin_err(twin_a, r["actor_grads"], r["actor_after"], "full-size actor step vs torch.optim on the same gradient") <= DISP_TOL, e
        assert twin_err(twin_c, r["critic_grads"], r["critic_after"], "full-size critic step vs torch.optim on the same gradient") <= DISP_TOL, e
        ap, cp = split(r["actor_after"], L.actor_spec), split(r["critic_after"], L.critic_spec)


def test_full_size_scan_linearity_and_padding_invariance():
    from cleanmarl_amd import _native as N
    lib = N.load()
    dev = torch.device("cuda:0")
    E, A, T = 4096, 8, 128
    g = torch.Generator().manual_seed(3)
    r1, r2 = torch.randn(E, T, generator=g).to(dev), torch.randn(E, T, generator=g).to(dev)
    v1, v2 = torch.randn(E, 1, T, generator=g).to(dev), torch.randn(E, 1, T, generator=g).to(dev)
    lens = torch.randint(1, T + 1, (E,), generator=g).int().to(dev)

    def scan(r, v):
        ret = torch.empty(E, A, T, device=dev); adv = torch.empty(E, A, T, device=dev)
        N.check(lib.cm_td_lambda_scan(N.ptr(r), N.ptr(v), N.ptr(lens), E, A, 1, T, 0.99, 0.95, N.ptr(ret), N.ptr(adv), N.stream_ptr()), "scan")
        return ret, adv
    ra, aa = scan(r1, v1); rb, ab = scan(r2, v2); rc, ac = scan(r1 + 2 * r2, v1 + 2 * v2)
    assert (rc - (ra + 2 * rb)).abs().max().item() <= 1e-4 and (ac - (aa + 2 * ab)).abs().max().item() <= 1e-4
    # garbage beyond ep_len must not leak into the valid region; padded outputs are exactly zero
    mask = torch.arange(T, device=dev)[None, :] < lens[:, None]
    r_g = torch.where(mask, r1, torch.full_like(r1, 1e6)); v_g = torch.where(mask[:, None, :], v1, torch.full_like(v1, -1e6))
    rg, ag = scan(r_g, v_g)
    assert torch.equal(rg, ra) and torch.equal(ag, aa)
    assert (ra * (~mask)[:, None, :]).abs().max().item() == 0.0 and (aa * (~mask)[:, None, :]).abs().max().item() == 0.0
    # agent broadcast: every agent sees the same sequence (MAPPO)
    assert torch.equal(ra[:, 0], ra[:, A - 1])


def test_full_size_update_is_deterministic_and_finite():
    L, b = _full_size_setup()
    r1 = L.train_iteration(b)
    L2, b2 = _full_size_setup()
    r2 = L2.train_iteration(b2)
    assert torch.equal(L.actor, L2.actor) and torch.equal(L.critic_params(), L2.critic_params())  # no atomics anywhere on the path
    assert all(np.isfinite(v) for r in r1 for v in r.values())
    assert [r["actor_loss"] for r in r1] == [r["actor_loss"] for r in r2]


# BASELINE.json configs[3] (IPPO, SMAClite-shaped: 2048 envs x 10 agents x 256 steps, obs 105 + 10 ids, 17 actions with availability masks,
# per-agent critic of width 32 as ippo_multienvs.py:34 defaults) and configs[1] (MAPPO 1024 x 3 x 128, obs 18 + 3 ids) at FULL size, in the
# storage the product uses (leading dimensions rounded up to 4 floats: the two-chunk k_mlp<2, ...> actor / k_mlp<-2> forward of config 4
# and the padded 21-wide rows of config 2 are otherwise only ever launched by bench.py)
# "cfg4_bench": the same config with the 2 x 64 critic that SURVEY.md 8's table and bench.py (Workload: cspec hidden 64) time -- the 64-wide
# k_critic_fused<2> over 5.2 M rows x 115 columns is 36 % of config 4's iteration and used to be launched at full size by bench.py only
_FULL = {"cfg4": dict(algo="ippo", E=2048, A=10, T=256, Do=115, Ds=243, K=17, Hc=32, avail_p=0.7, shard=16),
         "cfg4_bench": dict(algo="ippo", E=2048, A=10, T=256, Do=115, Ds=243, K=17, Hc=64, avail_p=0.7, shard=16),
         "cfg2": dict(algo="mappo", E=1024, A=3, T=128, Do=21, Ds=54, K=5, Hc=64, avail_p=1.0, shard=32),
         # config 3's shapes (the headline workload) for the tile-split / clock-probe test: 2^22 actor rows, one input chunk
         "cfg3_probe": dict(algo="mappo", E=4096, A=8, T=128, Do=56, Ds=384, K=5, Hc=64, avail_p=1.0, shard=32)}


def _full_size_cfg(name, seed=0):
    from cleanmarl_amd.learner import DeviceBatch, HParams, NetSpec, PPOLearner, init_params_like_torch
    c = _FULL[name]
    dev = torch.device("cuda:0")
    E, A, T, Do, Ds, K = c["E"], c["A"], c["T"], c["Do"], c["Ds"], c["K"]
    g = torch.Generator(device=dev).manual_seed(seed)
    b = DeviceBatch(E, A, T, Do, Ds, K, dev, pad_obs=True, pad_state=True)
    b.obs.copy_(torch.randn(E, A, T, Do, generator=g, device=dev)); b.state.copy_(torch.randn(E, T, Ds, generator=g, device=dev))
    av = torch.rand(E, A, T, K, generator=g, device=dev) < c["avail_p"]
    av[..., 0] = True  # action 0 is always legal
    b.avail.copy_(av.to(torch.uint8))
    # taken actions must be legal ones: the first legal action at or after a random index
    r = torch.randint(0, K, (E, A, T, 1), generator=g, device=dev)
    idx = (torch.arange(K, device=dev).view(1, 1, 1, K) + r) % K
    first = torch.gather(av, 3, idx).float().argmax(3, keepdim=True)
    b.action.copy_(torch.gather(idx, 3, first).squeeze(3).int())
    b.logp.copy_(-1.6 + 0.1 * torch.randn(E, A, T, generator=g, device=dev)); b.reward.copy_(torch.randn(E, T, generator=g, device=dev))
    b.ep_len.copy_(torch.randint(T // 2, T + 1, (E,), generator=g, device=dev).int())
    torch.manual_seed(1)
    aspec, cspec = NetSpec(Do, 64, 1, K), NetSpec(Ds if c["algo"] == "mappo" else Do, c["Hc"], 1, 1)
    L = PPOLearner(c["algo"], aspec, cspec, A, HParams(), dev, init_params_like_torch(aspec), init_params_like_torch(cspec))
    return L, b, c


@pytest.mark.parametrize("name", ["cfg4", "cfg4_bench", "cfg2"])
def test_full_size_other_configs_shard_additivity_and_oracle_shard(name):
    """Configs 4 and 2 at BASELINE size: the sums of uneven env shards add up to the full batch, N = b_mask.sum(), and one shard small
    enough for the CPU restatement (16 / 32 envs) matches it -- gradients of both networks, losses, returns and advantages."""
    from oracle import restatement as R
    from cleanmarl_amd import _native as N
    L, b, c = _full_size_cfg(name)
    algo, E = c["algo"], c["E"]
    L.compute_targets(b)
    s = N.stream_ptr()
    L.actor_pass(b, s); L.critic_pass(b, s)
    full = L.gbuf.clone()
    parts = torch.zeros_like(full)
    first = None
    n0 = c["shard"]
    for lo, hi in ((0, n0), (n0, E // 3 + 5), (E // 3 + 5, E)):
        sb = b.shard(lo, hi)
        L.actor_pass(sb, s); L.critic_pass(sb, s)
        parts += L.gbuf
        if first is None:
            first = L.gbuf.clone()
    torch.cuda.synchronize()
    assert (full - parts).abs().max().item() <= 1e-4 * full.abs().max().item()
    Pa, Pc = L.actor.numel(), L.critic.numel()
    n_rows = b.ep_len.sum().item()
    assert full[Pa + 5].item() == n_rows and full[Pa + 8 + Pc + 5].item() == n_rows  # N = b_mask.sum() for both networks (ippo_multienvs.py:571-572)
    sb = b.shard(0, n0)
    T = b.T
    mask = torch.arange(T)[None, :] < sb.ep_len.cpu()[:, None]
    batch = dict(obs=sb.obs.permute(0, 2, 1, 3).cpu(), actions=sb.action.permute(0, 2, 1).long().cpu(), log_probs=sb.logp.permute(0, 2, 1).cpu(),
                 reward=sb.reward.cpu(), states=sb.state.cpu(), avail=sb.avail.permute(0, 2, 1, 3).bool().cpu(), mask=mask)
    hp = dict(gamma=0.99, td_lambda=0.95, epochs=1, ppo_clip=0.2, entropy_coef=0.001, clip_gradients=-1, optimizer="Adam",
              learning_rate_actor=8e-4, learning_rate_critic=8e-4, normalize_reward=False, normalize_advantage=False, normalize_return=False)
    split = lambda flat, spec: [q.reshape(sh) for q, sh in zip(torch.split(flat.cpu(), [int(np.prod(sh)) for sh in spec.shapes()]), spec.shapes())]
    ap, cp = split(L.actor, L.actor_spec), split(L.critic_params(), L.critic_spec)
    ret, adv = R.prepare_targets(batch, cp, hp, algo)
    m3 = mask[:, :, None].numpy()
    assert _err(sb.ret.permute(0, 2, 1).cpu().numpy() * m3, ret.numpy() * m3) <= TOL
    assert _err(sb.adv.permute(0, 2, 1).cpu().numpy() * m3, adv.numpy() * m3) <= TOL
    scal, ag, cg = R.mlp_epoch(ap, cp, batch, sb.ret.permute(0, 2, 1).cpu(), sb.adv.permute(0, 2, 1).cpu(), hp, algo)
    n = float(first[Pa + 5])
    assert n == float(mask.sum())
    # gradients against the nearer of the fp32 and the fp64 evaluation of the restatement (tests/parity.py: KINK_GRAD_TOL)
    dbl = lambda v: v.double() if torch.is_tensor(v) and v.is_floating_point() else v
    _, ag64, cg64 = R.mlp_epoch([p.double() for p in ap], [p.double() for p in cp], {k: dbl(v) for k, v in batch.items()},
                                sb.ret.permute(0, 2, 1).cpu().double(), sb.adv.permute(0, 2, 1).cpu().double(), hp, algo)
    check_grads_kink(first[:Pa] / n, R.flat(ag), R.flat(ag64), "full-size shard actor grad")
    check_grads_kink(first[Pa + 8:Pa + 8 + Pc] / n, R.flat(cg), R.flat(cg64), "full-size shard critic grad")
    assert abs(float((-first[Pa + 0] - hp["entropy_coef"] * first[Pa + 1]) / n) - scal["actor_loss"]) <= TOL * (1 + abs(scal["actor_loss"]))
    assert abs(float(first[Pa + 8 + Pc + 4] / n) - scal["critic_loss"]) <= TOL * (1 + abs(scal["critic_loss"]))


@pytest.mark.parametrize("name", ["cfg4", "cfg4_bench", "cfg2"])
def test_full_size_other_configs_update_is_deterministic_and_finite(name):
    """Two independent full-size iterations (value pass, scan, three epochs on whichever schedule the size selects -- config 2 runs its
    critic epochs on the second stream) leave bit-identical parameters and finite records; the act pass over the whole episode
    (cm_policy_act_episode_ld: config 4's one-launch sampler) only ever picks legal actions and is reproducible."""
    from cleanmarl_amd import _native as N
    L, b, c = _full_size_cfg(name)
    r1 = L.train_iteration(b)
    L2, b2, _ = _full_size_cfg(name)
    r2 = L2.train_iteration(b2)
    assert torch.equal(L.actor, L2.actor) and torch.equal(L.critic_params(), L2.critic_params())
    assert all(np.isfinite(v) for r in r1 for v in r.values())
    assert [r["actor_loss"] for r in r1] == [r["actor_loss"] for r in r2] and [r["critic_loss"] for r in r1] == [r["critic_loss"] for r in r2]
    lib, dev, sp = N.load(), b.device, L.actor_spec
    outs = []
    for _ in range(2):
        act = torch.full((b.E, b.A, b.T), -1, dtype=torch.int32, device=dev); lp = torch.zeros(b.E, b.A, b.T, device=dev)
        w0b = lib.cm_w0_image_bytes(sp.din, sp.hidden)
        ws = torch.empty(max(w0b, 16), dtype=torch.uint8, device=dev)
        N.check(lib.cm_policy_act_episode_ld(N.ptr(b.obs), b.obs_ld, N.ptr(b.avail), b.E * b.A, b.T, sp.din, sp.hidden, sp.n_layers, sp.dout,
                                             N.ptr(L.actor), 1234, 0, N.ptr(act), N.ptr(lp), N.ptr(ws) if w0b else None, w0b, N.stream_ptr()),
                "cm_policy_act_episode_ld")
        outs.append((act, lp))
    torch.cuda.synchronize()
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    legal = torch.gather(b.avail, 3, outs[0][0].long().unsqueeze(3)).squeeze(3)
    assert bool(legal.all()) and bool(torch.isfinite(outs[0][1]).all()) and float(outs[0][1].max()) <= 1e-6


@pytest.mark.parametrize("algo,E,A,T,Do,Ds,K,H,L", [
    ("mappo", 1, 1, 1, 3, 5, 2, 8, 1),      # smallest possible batch: one env, one agent, one step
    ("ippo", 1, 2, 3, 4, 9, 1, 64, 1),      # single action (degenerate softmax), E = 1 (the reference's IPPO crashes here)
    ("mappo", 3, 4, 70, 130, 1100, 32, 64, 2),  # widest supported head / 3 actor chunks / split critic schedule, 3 column windows
    # the same corners on the layered schedule
    ("mappo", 1, 1, 1, 3, 5, 2, 65, 1),      # one row, one unit past the fused width (second 64-unit slab holds 1 unit)
    ("ippo", 1, 2, 3, 4, 9, 1, 128, 3),      # single action, E = 1, deep
    ("mappo", 3, 4, 70, 130, 1100, 32, 160, 1),  # 32 actions, 1100-wide critic input (3 column windows of the streaming dW), 2.5 slabs
])
def test_update_edge_shapes_match_oracle(algo, E, A, T, Do, Ds, K, H, L):
    _seeded_case(algo, E, A, T, Do, Ds, K, H, L, normalize=False)  # a single sample has no unbiased std


def test_all_steps_masked_is_finite_and_a_no_op_for_the_gradient():
    """ep_len == 0 everywhere (N = 0): statistics are zero, the gradient is zero, nothing is NaN."""
    from cleanmarl_amd import _native as N
    from cleanmarl_amd.learner import DeviceBatch, HParams, NetSpec, PPOLearner
    dev = torch.device("cuda:0")
    E, A, T, Do, Ds, K = 5, 3, 9, 21, 54, 5
    b = DeviceBatch(E, A, T, Do, Ds, K, dev)
    b.obs.normal_(); b.state.normal_(); b.avail.fill_(1); b.reward.normal_()  # ep_len stays 0
    torch.manual_seed(0)
    for H, nl in ((64, 1), (128, 3)):  # fused kernels, layered schedule
        L = PPOLearner("mappo", NetSpec(Do, H, nl, K), NetSpec(Ds, H, nl, 1), A, HParams(epochs=1), dev)
        L.compute_targets(b)
        assert (b.ret == 0).all() and (b.adv == 0).all()
        s = N.stream_ptr()
        L.actor_pass(b, s); L.critic_pass(b, s)
        torch.cuda.synchronize()
        assert torch.isfinite(L.gbuf).all() and (L.gbuf == 0).all()


def test_rollout_edge_shapes():
    """fused rollout with A that does not divide 64, a single env, T = 1; per-step path with an unsupported fused shape."""
    from cleanmarl_amd.learner import NetSpec, flatten_params, init_params_like_torch
    from cleanmarl_amd.rollout import SyntheticSpreadRollout
    dev = torch.device("cuda:0")
    torch.manual_seed(2)
    for (E, A, T) in [(1, 7, 1), (130, 9, 3), (5, 1, 4)]:
        ra = SyntheticSpreadRollout(E, A, T, seed=3, device=dev); rb = SyntheticSpreadRollout(E, A, T, seed=3, device=dev)
        spec = NetSpec(ra.Do, 64, 1, 5)
        p = flatten_params(init_params_like_torch(spec), dev)
        ba, bb = ra.collect(p, spec, fused=True), rb.collect(p, spec, fused=False)
        torch.cuda.synchronize()
        assert torch.equal(ba.obs[:, :, 0], bb.obs[:, :, 0])
        assert (ba.action == bb.action).float().mean().item() >= 0.98
        assert torch.isfinite(ba.reward).all() and torch.isfinite(ba.logp).all()
    r = SyntheticSpreadRollout(4, 12, 3, seed=3, device=dev)  # Do = 84 > 64: fused kernel refuses, per-step path serves it
    spec = NetSpec(r.Do, 64, 1, 5)
    p = flatten_params(init_params_like_torch(spec), dev)
    with pytest.raises(Exception):
        r.collect(p, spec, fused=True)
    b = r.collect(p, spec)
    torch.cuda.synchronize()
    assert torch.isfinite(b.obs).all() and (b.action >= 0).all() and (b.action < 5).all()


def test_time_padding_does_not_change_the_update():
    """pad_time (used to align T across env-sharded ranks) adds masked steps only: identical targets, losses, params.
    (MLP scripts only: the GRU scripts divide each TBPTT chunk loss by the chunk's step count INCLUDING padded steps,
    cleanmarl/mappo_lstm_multienvs.py:603-607, so there T must be -- and is -- the global maximum, as in the reference.)"""
    from cleanmarl_amd.learner import DeviceBatch, HParams, NetSpec, PPOLearner, init_params_like_torch, pad_time
    dev = torch.device("cuda:0")
    torch.manual_seed(4)
    batch = _random_case(77, 9, 3, 11, 21, 54, 5)
    aspec, cspec = NetSpec(21, 64, 1, 5), NetSpec(54, 64, 1, 1)
    ap, cp = init_params_like_torch(aspec), init_params_like_torch(cspec)
    outs = []
    for T_pad in (11, 20):
        b = DeviceBatch.from_reference_layout(batch["obs"], batch["actions"], batch["log_probs"], batch["reward"],
                                              batch["states"], batch["avail"], batch["mask"], dev)
        b = pad_time(b, T_pad)
        L = PPOLearner("mappo", aspec, cspec, 3, HParams(epochs=2), dev, [p.clone() for p in ap], [p.clone() for p in cp])
        recs = L.train_iteration(b)
        outs.append((b.ret[:, :, :11].clone(), L.actor.clone(), L.critic_params().clone(), [r["actor_loss"] for r in recs]))
    assert torch.equal(outs[0][0], outs[1][0])
    # row tiles are cut at different places (rows = (e*A + a)*T + t), so sums re-associate: equal to fp32 round-off
    assert _err(outs[0][1].cpu().numpy(), outs[1][1].cpu().numpy()) <= 1e-6
    assert _err(outs[0][2].cpu().numpy(), outs[1][2].cpu().numpy()) <= 1e-6
    assert _err(outs[0][3], outs[1][3]) <= 1e-6


def test_shape_env_kernels_match_numpy_twin():
    """cm_shape_env_fill / cm_shape_env_reward vs cleanmarl_amd/env/synthetic.py::SyntheticShapeEnv."""
    from cleanmarl_amd.env.synthetic import SyntheticShapeEnv
    from cleanmarl_amd.learner import NetSpec, flatten_params, init_params_like_torch
    from cleanmarl_amd.rollout import SyntheticShapeRollout
    dev = torch.device("cuda:0")
    torch.manual_seed(8)
    E, A, T, raw, Ds, K = 6, 4, 5, 13, 9, 7
    r = SyntheticShapeRollout(E, A, T, obs_raw=raw, state_dim=Ds, n_actions=K, avail_p=0.6, seed=11, device=dev, env_offset=40)
    spec = NetSpec(r.Do, 64, 1, K)
    p = flatten_params(init_params_like_torch(spec), dev)
    for ep in range(2):
        b = r.collect(p, spec)
    torch.cuda.synchronize()
    obs, st, av, act, rew = (x.cpu().numpy() for x in (b.obs, b.state, b.avail, b.action, b.reward))
    assert av[..., 0].all() and 0.3 < av[..., 1:].mean() < 0.9
    assert (av[np.arange(E)[:, None, None], np.arange(A)[None, :, None], np.arange(T)[None, None, :], act] == 1).all()
    for e in range(E):
        env = SyntheticShapeEnv(A, raw, Ds, K, 0.6, True, T, seed=11, env_index=40 + e)
        env.episode = 0  # the next reset() is episode 1 = the second collect()
        o, _ = env.reset()
        for t in range(T):
            assert np.abs(obs[e, :, t] - o).max() <= 1e-5 and np.abs(st[e, t] - env.get_state()).max() <= 1e-5
            assert (av[e, :, t] == env.get_avail_actions()).all()
            o, rr, d, tr, _ = env.step(act[e, :, t])
            assert abs(rew[e, t] - rr) <= 1e-5


def test_episode_act_equals_per_step_act():
    """cm_policy_act_episode (one launch for all T steps) == T x cm_policy_act on the shape env."""
    from cleanmarl_amd.learner import NetSpec, flatten_params, init_params_like_torch
    from cleanmarl_amd.rollout import SyntheticShapeRollout
    dev = torch.device("cuda:0")
    torch.manual_seed(12)
    ra = SyntheticShapeRollout(9, 3, 7, obs_raw=29, state_dim=11, n_actions=17, seed=5, device=dev, env_offset=3)
    rb = SyntheticShapeRollout(9, 3, 7, obs_raw=29, state_dim=11, n_actions=17, seed=5, device=dev, env_offset=3)
    spec = NetSpec(ra.Do, 64, 1, 17)
    p = flatten_params(init_params_like_torch(spec), dev)
    ba, bb = ra.collect(p, spec, fused=True), rb.collect(p, spec, fused=False)
    torch.cuda.synchronize()
    assert torch.equal(ba.action, bb.action) and torch.equal(ba.logp, bb.logp) and torch.equal(ba.reward, bb.reward)


def test_policy_act_greedy_is_the_argmax_of_the_masked_logits():
    from oracle import restatement as R
    from cleanmarl_amd import _native as N
    from cleanmarl_amd.learner import NetSpec, flatten_params, init_params_like_torch
    lib, dev = N.load(), torch.device("cuda:0")
    torch.manual_seed(3)
    for rows, Do, K, L in [(777, 30, 9, 1), (65, 70, 5, 2)]:
        spec = NetSpec(Do, 64, L, K)
        p = init_params_like_torch(spec)
        obs = torch.randn(rows, Do)
        avail = torch.rand(rows, K) < 0.5
        avail[:, 1] = True
        action = torch.empty(rows, dtype=torch.int32, device=dev); logp = torch.empty(rows, device=dev)
        d_obs, d_av, d_p = obs.to(dev), avail.to(torch.uint8).to(dev), flatten_params(p, dev)
        N.check(lib.cm_policy_act_greedy(N.ptr(d_obs), Do, N.ptr(d_av), K, rows, Do, 64, L, K, N.ptr(d_p), N.ptr(action), N.ptr(logp), 1,
                                         N.stream_ptr()), "greedy")
        logits = R.actor_logits(p, obs, avail)
        lp_all = torch.log_softmax(logits, -1)
        a = action.cpu().long()
        top = logits.max(-1).values
        assert (logits.gather(-1, a[:, None])[:, 0] >= top - 1e-5).all()  # a maximiser (ties within round-off allowed)
        assert avail[torch.arange(rows), a].all()
        assert _err(logp.cpu().numpy(), lp_all.gather(-1, a[:, None])[:, 0].numpy()) <= TOL


@pytest.mark.parametrize("kind", ["Adam", "AdamW", "SGD", "RMSprop"])
def test_optimiser_step_matches_torch_at_size_boundaries(kind):
    """cm_grad_norm_clip_adam against torch.optim + clip_grad_norm_ (cleanmarl/mappo_multienvs.py:584-594) on sizes around the switch
    between the two-launch form (<= 32768 parameters: norm pass + multi-workgroup update) and the one-workgroup kernel, with the
    gradient scaled by grad_scale / N as the learners pass it, with and without clipping, three consecutive steps."""
    from cleanmarl_amd import _native as N
    lib = N.load()
    dev = torch.device("cuda:0")
    kinds = {"Adam": N.OPT_ADAM, "AdamW": N.OPT_ADAMW, "SGD": N.OPT_SGD, "RMSprop": N.OPT_RMSPROP}
    for n in (1, 1000, 1024, 4097, 32767, 32768, 32769, 100003):
        for max_norm in (-1.0, 0.5):
            gen = torch.Generator().manual_seed(n)
            p0 = torch.randn(n, generator=gen)
            ref = torch.nn.Parameter(p0.clone())
            kw = dict(lr=8e-4)
            opt = getattr(torch.optim, kind)([ref], **kw)
            p = p0.clone().to(dev)
            m, v = torch.zeros(n, device=dev), torch.zeros(n, device=dev)
            norm = torch.zeros(1, device=dev)
            count, grad_scale = 37.0, 0.25
            for step in (1, 2, 3):
                gsum = torch.randn(n, generator=gen) * 50.0
                ref.grad = gsum * (grad_scale / count)
                want_norm = float(ref.grad.norm())
                if max_norm > 0:
                    torch.nn.utils.clip_grad_norm_([ref], max_norm)
                want_grad = ref.grad.clone()
                opt.step()
                g = torch.zeros(n + N.NUM_STATS, device=dev)
                g[:n] = gsum.to(dev)
                g[n + N.STAT_COUNT] = count
                N.check(lib.cm_grad_norm_clip_adam(N.ptr(p), N.ptr(g), N.ptr(m), N.ptr(v), n, step, 8e-4, 0.9,
                                                   0.99 if kind == "RMSprop" else 0.999, 1e-8, 0.01 if kind == "AdamW" else 0.0,
                                                   kinds[kind], max_norm, grad_scale, N.ptr(norm), N.stream_ptr()), "cm_grad_norm_clip_adam")
                torch.cuda.synchronize()
                assert abs(float(norm) - want_norm) <= 1e-5 * (1 + want_norm), (kind, n, step)
                assert _err(g[:n].cpu().numpy(), want_grad.numpy()) <= 1e-6, (kind, n, max_norm, step)  # what optimizer.step() consumed
                assert _err(p.cpu().numpy(), ref.detach().numpy()) <= 1e-6, (kind, n, max_norm, step)


@pytest.mark.parametrize("algo,E,A,T,Do,Ds,K,H,L", [("mappo", 37, 3, 25, 21, 54, 5, 64, 1), ("ippo", 12, 10, 40, 115, 243, 17, 64, 1),
                                                      ("mappo", 9, 5, 12, 35, 150, 5, 64, 1), ("mappo", 7, 2, 9, 131, 390, 6, 48, 2)])
def test_padded_leading_dimensions_do_not_change_the_update(algo, E, A, T, Do, Ds, K, H, L):
    """The `_ld` entry points (obs / state rows padded to a multiple of 4 floats so that tile loads are 16-byte wide; feature widths
    21 / 35 / 115 of BASELINE configs 2 / 5 / 4, and 131 / 390 = three chunks + the split critic schedule): same returns, losses,
    gradients and post-step parameters as the unpadded batch, and both within 1e-4 of the oracle."""
    from oracle import restatement as R
    from cleanmarl_amd.learner import DeviceBatch, HParams, NetSpec, PPOLearner, init_params_like_torch
    dev = torch.device("cuda:0")
    torch.manual_seed(5)
    batch = _random_case(31, E, A, T, Do, Ds, K)
    aspec, cspec = NetSpec(Do, H, L, K), NetSpec(Ds if algo == "mappo" else Do, H, L, 1)
    ap, cp = init_params_like_torch(aspec), init_params_like_torch(cspec)
    hpd = dict(gamma=0.99, td_lambda=0.95, normalize_advantage=True, normalize_return=False, epochs=2, ppo_clip=0.2, entropy_coef=0.01,
               clip_gradients=0.5, optimizer="Adam", learning_rate_actor=8e-4, learning_rate_critic=8e-4)
    recs = []
    for pad in (False, True):
        b = DeviceBatch.from_reference_layout(batch["obs"], batch["actions"], batch["log_probs"], batch["reward"], batch["states"],
                                              batch["avail"], batch["mask"], dev, pad=pad)
        assert (b.obs_ld % 4 == 0 and b.state_ld % 4 == 0) if pad else (b.obs_ld == Do and b.state_ld == Ds)
        L_ = PPOLearner(algo, aspec, cspec, A, HParams(**hpd), dev, [p.clone() for p in ap], [p.clone() for p in cp])
        r = L_.train_iteration(b, keep_grads=True)
        torch.cuda.synchronize()
        recs.append((b.ret.clone(), b.adv.clone(), [dict(d) for d in r]))
    (ret0, adv0, r0), (ret1, adv1, r1) = recs
    assert torch.equal(ret0, ret1) and torch.equal(adv0, adv1)  # the value pass reads the same numbers in the same order
    ret, adv, orec = R.mlp_update(ap, cp, batch, hpd, algo)
    assert _err(ret1.permute(0, 2, 1).cpu().numpy(), ret.numpy()) <= TOL
    for e in range(2):
        for k in ("actor_loss", "critic_loss", "entropy", "kl", "clipfrac", "actor_gnorm", "critic_gnorm"):
            assert abs(r0[e][k] - r1[e][k]) <= 1e-6 * (1 + abs(r0[e][k])), (k, r0[e][k], r1[e][k])
            assert abs(r1[e][k] - orec[e][k]) <= TOL * (1 + abs(orec[e][k])), (k, r1[e][k], orec[e][k])
        for k in ("actor_grads", "critic_grads", "actor_after", "critic_after"):
            assert _err(r0[e][k].cpu().numpy(), r1[e][k].cpu().numpy()) <= 1e-6, k
            assert _err(r1[e][k].cpu().numpy(), R.flat(orec[e][k]).numpy()) <= TOL, k


@pytest.mark.parametrize("algo,E,A,T,Do,Ds,K,H,L", [("mappo", 37, 3, 25, 21, 54, 5, 64, 1), ("mappo", 13, 8, 20, 56, 384, 5, 48, 1)])
def test_hand_ordered_and_compiler_scheduled_product_forms_agree(algo, E, A, T, Do, Ds, K, H, L, monkeypatch):
    """k_mlp's 32x32x2 products exist in two forms (csrc/cm_mlp_kernel.h: LDS reads issued by hand through inline asm, or left to the
    compiler); launches pick one by shape and size.  Same operands, same order: the targets and the critic's whole update must be identical
    bits with either.  The ACTOR pass of the hand-ordered instantiation also carries the 4x4x1-MFMA head of round 6 (other summation order in
    the three head products): its gradients agree with the compiler-scheduled twin's to 1e-5 of their largest entry instead of bit for bit, and
    both sit within the bars of the oracle.  (The hand-ordered actor pass only runs from 2^21 rows on by default: forced here.)"""
    from oracle import restatement as R
    from cleanmarl_amd.learner import DeviceBatch, HParams, NetSpec, PPOLearner, init_params_like_torch
    dev = torch.device("cuda:0")
    torch.manual_seed(7)
    batch = _random_case(77, E, A, T, Do, Ds, K)
    aspec, cspec = NetSpec(Do, H, L, K), NetSpec(Ds, H, L, 1)
    ap, cp = init_params_like_torch(aspec), init_params_like_torch(cspec)
    hpd = dict(gamma=0.99, td_lambda=0.95, normalize_advantage=True, normalize_return=False, epochs=2, ppo_clip=0.2, entropy_coef=0.01,
               clip_gradients=0.5, optimizer="Adam", learning_rate_actor=8e-4, learning_rate_critic=8e-4)
    out = {}
    for forms in ("hand", "loop"):
        monkeypatch.setenv("CM_MLP_FORMS", forms)
        b = DeviceBatch.from_reference_layout(batch["obs"], batch["actions"], batch["log_probs"], batch["reward"], batch["states"],
                                              batch["avail"], batch["mask"], dev, pad=True)  # 16-byte rows: the forms' common shape
        L_ = PPOLearner(algo, aspec, cspec, A, HParams(**hpd), dev, [p.clone() for p in ap], [p.clone() for p in cp])
        r = L_.train_iteration(b, keep_grads=True)
        torch.cuda.synchronize()
        out[forms] = (b.ret.clone(), [dict(d) for d in r], L_.actor.clone(), L_.critic_params().clone())
    assert torch.equal(out["hand"][0], out["loop"][0])
    assert torch.equal(out["hand"][3], out["loop"][3])
    for e in range(2):
        assert torch.equal(out["hand"][1][e]["critic_grads"], out["loop"][1][e]["critic_grads"])
    g0h, g0l = out["hand"][1][0]["actor_grads"], out["loop"][1][0]["actor_grads"]
    assert not torch.equal(g0h, g0l), "the hand-ordered actor pass did not take its own head"
    assert grad_err(g0h, g0l, "hand vs loop forms, actor gradient") <= 1e-5
    assert _err(out["hand"][2].cpu().numpy(), out["loop"][2].cpu().numpy(), "hand vs loop forms, actor parameters") <= 1e-5
    before = R.flat(ap).clone()  # (before mlp_update steps the lists in place)
    ret, adv, orec = R.mlp_update(ap, cp, batch, hpd, algo)
    assert _err(out["hand"][0].permute(0, 2, 1).cpu().numpy(), ret.numpy()) <= TOL
    for forms in ("hand", "loop"):
        ca = StepChecker(before, "Adam", 8e-4, forms + " forms actor")
        for e in range(2):
            ca.step(out[forms][1][e]["actor_grads"], out[forms][1][e]["actor_after"], R.flat(orec[e]["actor_grads"]), R.flat(orec[e]["actor_after"]))


@pytest.mark.parametrize("E,A,T,Do,K,H", [(37, 3, 25, 20, 2, 64), (13, 8, 20, 56, 5, 48), (29, 2, 33, 12, 8, 17), (64, 4, 16, 24, 3, 64), (5, 1, 7, 8, 7, 33)])
def test_the_4x4_mfma_head_matches_the_oracle_on_odd_shapes(E, A, T, Do, K, H, monkeypatch):
    """The wave-private head of the hand-ordered actor pass (csrc/cm_mlp_kernel.h: head_logits44 / head_bwd_wave44 on v_mfma_f32_4x4x1_16b_f32) against the
    oracle's whole update: 2 .. 8 actions (padding columns and the 4-output groups), hidden widths below 64 (zero-padded operand columns), row counts with
    partial tiles, ragged episodes, unavailable actions -- every bar of the seeded cases (gradient 1e-4 of its largest entry, the optimiser twin, 1e-4
    end to end).  Forced with CM_MLP_FORMS=hand: by default these sizes run the compiler-scheduled twin."""
    monkeypatch.setenv("CM_MLP_FORMS", "hand")
    _seeded_case("mappo", E, A, T, Do, 6 * A * A, K, H, 1, normalize=True, pad=True)


def test_one_pass_critic_equals_the_two_kernel_schedule_on_random_shapes(monkeypatch):
    """cm_critic_fwd_bwd_ld straight through the C-ABI, 24 seeded random shapes inside the one-pass kernel's domain (2 .. 7 input chunks,
    H <= 64, central and per-agent targets, ragged episode lengths, row counts with partial tiles and more tiles than workgroups, padded and
    exact leading dimensions): gradient + statistics buffer of CM_CRITIC_SCHEDULE=fused against =split (both against the oracle elsewhere;
    here the two schedules against each other, 1e-4 of the buffer's scale)."""
    from cleanmarl_amd import _native as N
    from cleanmarl_amd.learner import NetSpec, flatten_params, init_params_like_torch
    lib, dev = N.load(), torch.device("cuda:0")
    rng = np.random.default_rng(2024)
    for case in range(24):
        nc = int(rng.integers(2, 8))
        din = int(rng.integers(64 * (nc - 1) + 1, 64 * nc + 1))
        ld = (din + 3) // 4 * 4 + 4 * int(rng.integers(0, 3))
        H = int(rng.choice([16, 33, 48, 64]))
        A, T = int(rng.integers(1, 12)), int(rng.integers(3, 40))
        per_agent = int(rng.integers(0, 2))
        E = int(rng.choice([1, 3, 17, 70, 700 if case % 6 == 0 else 40]))
        torch.manual_seed(case)
        spec = NetSpec(din, H, 1, 1)
        p = flatten_params(init_params_like_torch(spec), dev)
        rows_shape = (E, A, T) if per_agent else (E, T)
        x = torch.zeros(*rows_shape, ld, device=dev)
        x[..., :din] = torch.randn(*rows_shape, din, device=dev)
        ret = torch.randn(E, A, T, device=dev)
        ep_len = torch.from_numpy(rng.integers(1, T + 1, size=E).astype(np.int32)).to(dev)
        ws = torch.empty(lib.cm_critic_workspace_bytes(E, A, T, per_agent, din, H, 1), dtype=torch.uint8, device=dev)
        out = {}
        for sched in ("fused", "split", "fused2"):  # fused2: two row tiles per iteration for two-chunk inputs (k_critic_fused2, opt-in)
            N.set_option("critic_schedule", sched)
            g = torch.full((spec.nparams + N.NUM_STATS,), float("nan"), device=dev)
            N.check(lib.cm_critic_fwd_bwd_ld(N.ptr(x), ld, N.ptr(ret), N.ptr(ep_len), E, A, T, per_agent, din, H, 1, N.ptr(p), N.ptr(g),
                                             N.ptr(ws), ws.numel(), N.stream_ptr()), "cm_critic_fwd_bwd_ld")
            torch.cuda.synchronize()
            out[sched] = g.cpu().numpy()
        assert np.isfinite(out["fused"]).all(), (case, din, H, A, T, E, per_agent)
        scale = 1.0 + np.abs(out["split"]).max()
        assert np.abs(out["fused"] - out["split"]).max() <= TOL * scale, (case, din, ld, H, A, T, E, per_agent)
        assert np.abs(out["fused"] - out["fused2"]).max() <= 1e-5 * scale, (case, din, ld, H, A, T, E, per_agent)
        if nc != 2:
            assert np.array_equal(out["fused"], out["fused2"])  # only two-chunk inputs have a two-tile form


def test_first_critic_epoch_on_the_value_pass_activations_matches_the_plain_pass():
    """cm_value_pass_keep_h0_ld + cm_critic_fwd_bwd_h0_ld straight through the C-ABI, 24 seeded random shapes of the one-pass critic (2 .. 7 input chunks,
    H <= 64, central and per-agent targets, ragged episodes, partial tiles, more tiles than workgroups, padded leading dimensions): the values are the
    bits of cm_mlp_forward_solo_ld, h0 is relu(x W0^T + b0) (1e-5 against torch), and the critic pass that reads h0 instead of multiplying by W0
    (k_critic_fused<NC, true>) returns the gradient + statistics buffer of the plain one-pass kernel to 1e-5 of its scale (h0 comes from another
    product form: not bit for bit), with identical row count."""
    from cleanmarl_amd import _native as N
    from cleanmarl_amd.learner import NetSpec, flatten_params, init_params_like_torch
    lib, dev = N.load(), torch.device("cuda:0")
    rng = np.random.default_rng(777)
    N.set_option("critic_schedule", "fused")
    try:
        for case in range(24):
            nc = int(rng.integers(2, 8))
            din = int(rng.integers(64 * (nc - 1) + 1, 64 * nc + 1))
            ld = (din + 3) // 4 * 4 + 4 * int(rng.integers(0, 3))
            H = int(rng.choice([16, 33, 48, 64]))
            A, T = int(rng.integers(1, 12)), int(rng.integers(3, 40))
            per_agent = int(rng.integers(0, 2))
            E = int(rng.choice([1, 3, 17, 70, 700 if case % 6 == 0 else 40]))
            torch.manual_seed(case)
            spec = NetSpec(din, H, 1, 1)
            params = init_params_like_torch(spec)
            p = flatten_params(params, dev)
            rows_shape = (E, A, T) if per_agent else (E, T)
            rows = int(np.prod(rows_shape))
            x = torch.zeros(*rows_shape, ld, device=dev)
            x[..., :din] = torch.randn(*rows_shape, din, device=dev)
            ret = torch.randn(E, A, T, device=dev)
            ep_len = torch.from_numpy(rng.integers(1, T + 1, size=E).astype(np.int32)).to(dev)
            ws = torch.empty(max(lib.cm_critic_workspace_bytes(E, A, T, per_agent, din, H, 1), lib.cm_mlp_forward_workspace_bytes(rows, din, H, 1, 1)),
                             dtype=torch.uint8, device=dev)
            v0, v1 = torch.empty(rows, device=dev), torch.empty(rows, device=dev)
            h0 = torch.full((rows, 64), float("nan"), device=dev)
            N.check(lib.cm_mlp_forward_solo_ld(N.ptr(x), ld, rows, din, H, 1, 1, N.ptr(p), None, N.ptr(v0), N.ptr(ws), ws.numel(), N.stream_ptr()), "solo")
            N.check(lib.cm_value_pass_keep_h0_ld(N.ptr(x), ld, rows, din, H, 1, N.ptr(p), N.ptr(v1), N.ptr(h0), N.ptr(ws), ws.numel(), N.stream_ptr()), "keep_h0")
            torch.cuda.synchronize()
            assert torch.equal(v0, v1), case
            W0, b0 = params[0].to(dev), params[1].to(dev)
            h_ref = torch.relu(x.reshape(rows, ld)[:, :din].double() @ W0.double().T + b0.double()).float()
            assert _err(h0[:, :H].cpu().numpy(), h_ref.cpu().numpy(), "value pass h0") <= 1e-5, case
            assert (h0[:, H:] == 0).all(), case
            out = {}
            for kind in ("plain", "h0"):
                g = torch.full((spec.nparams + N.NUM_STATS,), float("nan"), device=dev)
                if kind == "plain":
                    N.check(lib.cm_critic_fwd_bwd_ld(N.ptr(x), ld, N.ptr(ret), N.ptr(ep_len), E, A, T, per_agent, din, H, 1, N.ptr(p), N.ptr(g),
                                                     N.ptr(ws), ws.numel(), N.stream_ptr()), "cm_critic_fwd_bwd_ld")
                else:
                    N.check(lib.cm_critic_fwd_bwd_h0_ld(N.ptr(x), ld, N.ptr(h0), N.ptr(ret), N.ptr(ep_len), E, A, T, per_agent, din, H, 1, N.ptr(p), N.ptr(g),
                                                        N.ptr(ws), ws.numel(), N.stream_ptr()), "cm_critic_fwd_bwd_h0_ld")
                torch.cuda.synchronize()
                out[kind] = g.cpu().numpy()
            assert np.isfinite(out["h0"]).all(), (case, din, H, A, T, E, per_agent)
            scale = 1.0 + np.abs(out["plain"]).max()
            assert np.abs(out["h0"] - out["plain"]).max() <= 1e-5 * scale, (case, din, ld, H, A, T, E, per_agent, np.abs(out["h0"] - out["plain"]).max() / scale)
            assert out["h0"][spec.nparams + N.STAT_COUNT] == out["plain"][spec.nparams + N.STAT_COUNT], case
    finally:
        N.set_option("critic_schedule", "auto")


@pytest.mark.parametrize("per_agent", [0, 1])
@pytest.mark.parametrize("rows_e,T", [(2, 64), (3, 64), (5, 40), (33, 31), (1030, 64), (2049, 33)])
def test_two_tile_critic_equals_the_one_tile_kernel(rows_e, T, per_agent):
    """k_critic_fused2 (critic_schedule = "fused2", opt-in: two row tiles per iteration, two-chunk inputs -- config 4's per-agent critic on 115-wide observations) against
    k_critic_fused on tile counts that are even, odd (the last pair's second tile lies past the end), partial, fewer and more than two per
    workgroup: the same sums up to the association of a workgroup's tiles (1e-5 of the buffer's scale), equal N, two launches bit-identical."""
    from cleanmarl_amd import _native as N
    from cleanmarl_amd.learner import NetSpec, flatten_params, init_params_like_torch
    lib, dev = N.load(), torch.device("cuda:0")
    din, ld, H, A = 115, 116, 64, 3
    E = rows_e
    torch.manual_seed(rows_e)
    spec = NetSpec(din, H, 1, 1)
    p = flatten_params(init_params_like_torch(spec), dev)
    shape = (E, A, T) if per_agent else (E, T)
    x = torch.zeros(*shape, ld, device=dev)
    x[..., :din] = torch.randn(*shape, din, device=dev)
    ret = torch.randn(E, A, T, device=dev)
    g0 = torch.Generator().manual_seed(rows_e)
    ep_len = torch.randint(1, T + 1, (E,), generator=g0).int().to(dev)
    ws = torch.empty(lib.cm_critic_workspace_bytes(E, A, T, per_agent, din, H, 1), dtype=torch.uint8, device=dev)
    out = {}
    try:
        for sched in ("fused2", "fused2", "fused"):
            N.set_option("critic_schedule", sched)
            g = torch.full((spec.nparams + N.NUM_STATS,), float("nan"), device=dev)
            N.check(lib.cm_critic_fwd_bwd_ld(N.ptr(x), ld, N.ptr(ret), N.ptr(ep_len), E, A, T, per_agent, din, H, 1, N.ptr(p), N.ptr(g),
                                             N.ptr(ws), ws.numel(), N.stream_ptr()), "cm_critic_fwd_bwd_ld")
            torch.cuda.synchronize()
            out.setdefault(sched, []).append(g.cpu().numpy())
    finally:
        N.set_option("critic_schedule", "auto")
    two, again, one = out["fused2"][0], out["fused2"][1], out["fused"][0]
    assert np.isfinite(two).all() and np.array_equal(two, again)
    P = spec.nparams
    assert two[P + N.STAT_COUNT] == one[P + N.STAT_COUNT] == float(ep_len.sum())
    assert np.abs(two - one).max() <= 1e-5 * (1.0 + np.abs(one).max())


def test_padded_rollout_buffers_hold_the_same_rollout():
    """cm_rollout_spread_ld / cm_shape_env_fill_ld / cm_policy_act_episode_ld: buffers with padded leading dimensions receive exactly the
    rollout of the unpadded ones (bit for bit), and their padding columns stay zero."""
    from cleanmarl_amd.learner import NetSpec, flatten_params, init_params_like_torch
    from cleanmarl_amd.rollout import SyntheticShapeRollout, SyntheticSpreadRollout
    dev = torch.device("cuda:0")
    torch.manual_seed(9)
    for mk, spec_k in ((lambda pad: SyntheticSpreadRollout(50, 3, 9, seed=7, device=dev, env_offset=11, pad=pad), 5),
                       (lambda pad: SyntheticSpreadRollout(600, 5, 6, seed=7, device=dev, env_offset=11, pad=pad), 5),
                       (lambda pad: SyntheticShapeRollout(9, 3, 7, obs_raw=29, state_dim=11, n_actions=17, seed=5, device=dev, env_offset=3, pad=pad), 17)):
        ra, rb = mk(False), mk(True)
        spec = NetSpec(ra.Do, 64, 1, spec_k)
        p = flatten_params(init_params_like_torch(spec), dev)
        for _ in range(2):
            ba, bb = ra.collect(p, spec), rb.collect(p, spec)
        torch.cuda.synchronize()
        assert bb.obs_ld % 4 == 0 and bb.state_ld % 4 == 0 and (bb.obs_ld != ra.Do or bb.state_ld != ra.Ds)
        for k in ("obs", "state", "avail", "action", "logp", "reward"):
            assert torch.equal(getattr(ba, k), getattr(bb, k)), k
        st = bb.obs.as_strided((bb.E, bb.A, bb.T, bb.obs_ld), (bb.A * bb.T * bb.obs_ld, bb.T * bb.obs_ld, bb.obs_ld, 1))
        assert not st[..., bb.Do:].any()


def test_unequal_tile_split_and_clock_probe_at_full_size():
    """Round 5: (i) the unequal static tile split of full-size launches (cm_mlp_kernel.h::set_tile_split: the first half of a two-per-CU
    grid takes 56 % of the tiles) is a pure re-mapping of tiles to workgroups -- the value pass (rows independent) is BIT-identical to
    the equal split, the actor pass's folded gradient differs by fp32 re-association only (different partial sums, <= 1e-5 relative) and is
    itself run-to-run identical; (ii) cm_clock_probe: every workgroup of the actor pass reports where and how fast it ran."""
    from cleanmarl_amd import _native as N
    L, b, c = _full_size_cfg("cfg3_probe")
    lib, s = N.load(), N.stream_ptr()
    L.compute_targets(b)
    outs = {}
    try:
        for split in ("50", "auto", "auto"):
            N.set_option("tile_split", split)
            L.compute_targets(b)  # value pass + scan (forward kernel, 2^19 rows: split because it runs alone)
            clk = torch.zeros(512, 4, dtype=torch.int64, device=b.device)
            N.check(lib.cm_clock_probe(N.ptr(clk)), "cm_clock_probe")
            L.actor_pass(b, s)
            torch.cuda.synchronize()
            N.check(lib.cm_clock_probe(None), "cm_clock_probe")
            outs.setdefault(split, []).append((L.values.clone(), L.g_actor.clone(), clk.cpu()))
    finally:
        N.set_option("tile_split", "auto")
    (v50, g50, _), = outs["50"]
    (va, ga, clk), (vb, gb, _) = outs["auto"]
    assert torch.equal(v50, va) and torch.equal(va, vb)          # forward: same bits whatever the tile -> workgroup map
    assert torch.equal(ga, gb)                                   # the split is static: run-to-run identical
    Pa = L.actor.numel()
    assert not torch.equal(ga[:Pa], g50[:Pa])                    # it IS a different association of the partial sums ...
    assert (ga[:Pa] - g50[:Pa]).abs().max().item() <= 1e-5 * g50[:Pa].abs().max().item()   # ... and nothing more
    assert ga[Pa + N.STAT_COUNT].item() == g50[Pa + N.STAT_COUNT].item() == b.ep_len.sum().item()
    # the probe: 512 workgroups, two per CU on 256 distinct CUs of 8 XCDs, a plausible shader clock, sane stamps
    live = clk[clk[:, 3] > 0]
    assert live.shape[0] == 512
    xcc = (live[:, 1] >> 32) & 0xF
    assert sorted(set(xcc.tolist())) == list(range(8))
    ghz = live[:, 0].double() / ((live[:, 3] - live[:, 2]).double() / 1e8) / 1e9
    assert 1.0 < float(ghz.min()) and float(ghz.max()) < 2.7, (float(ghz.min()), float(ghz.max()))
    span = (live[:, 3].max() - live[:, 2].min()).item() / 1e5   # ms
    assert 0.5 < span < 10.0
