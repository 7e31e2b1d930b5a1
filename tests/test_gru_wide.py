"""GPU tests of the LAYERED GRU schedule (csrc/cm_gru_wide.hip): recurrent actors on observations wider than 64 columns and / or with
65..256 hidden units -- shapes the reference's Actor accepts (cleanmarl/mappo_lstm_multienvs.py:162-184: any input_dim / hidden_dim;
SMAClite observations are ~100+ wide, a 10-agent MPE 70) and the fused sweeps do not hold.  Same oracle, same 1e-4 bar, same
entry points as the fused shapes (cm_gru_actor_chunk_fwd_bwd / _train_step, cm_gru_policy_act_ws)."""
import math

import numpy as np
import pytest
import torch

from test_hip_parity import _check_gru_against_oracle, _random_case
from parity import DISP_TOL, GRAD_TOL, TOL, check_grads, check_step, golden_before, golden_init, grad_err  # noqa: F401
from parity import err as _err

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("fused_step", [True, False])
@pytest.mark.parametrize("algo,E,A,T,Do,Ds,K,H,tb", [
    ("mappo", 9, 10, 11, 70, 60, 5, 64, 4),      # 10-agent MPE observation width, 64 hidden units (only the input is wide)
    ("ippo", 7, 3, 13, 115, 50, 17, 64, 5),      # BASELINE config 4's observation width and action count
    ("mappo", 6, 3, 10, 21, 54, 5, 96, 4),       # narrow input, 1.5 slabs of hidden units
    ("ippo", 5, 4, 12, 115, 40, 17, 128, 5),     # both: SMAClite-shaped obs on a 128-wide GRU
    ("mappo", 130, 2, 7, 70, 30, 6, 128, 7),     # several row tiles, one chunk
    ("mappo", 4, 2, 9, 33, 20, 4, 200, 3),       # more than three 64-unit slabs
    ("ippo", 7, 3, 13, 35, 50, 36, 64, 5),       # a fused-size GRU whose head has more than 32 actions (64-wide logits plane)
    ("mappo", 5, 2, 9, 70, 30, 64, 96, 4),       # widest head
])
def test_layered_gru_update_matches_oracle(algo, E, A, T, Do, Ds, K, H, tb, fused_step):
    from oracle import restatement as R
    from cleanmarl_amd.gru import GRUPPOLearner
    from cleanmarl_amd.learner import DeviceBatch, HParams, NetSpec, init_params_like_torch
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
    L.fused_step = fused_step
    recs = L.train_iteration(b, keep_grads=True)
    a0, c0 = R.flat(ap).clone(), R.flat(cp).clone()  # (gru_update steps the lists in place)
    ret, adv, orecs = R.gru_update(ap, cp, batch, hp, algo)
    assert _err(b.ret.permute(0, 2, 1).cpu().numpy(), ret.numpy()) <= TOL and _err(b.adv.permute(0, 2, 1).cpu().numpy(), adv.numpy()) <= TOL
    _check_gru_against_oracle(recs, orecs, a0, c0, hp, "gru layered")


@pytest.mark.parametrize("rows,Do,Hd,K", [(150, 115, 64, 17), (70, 35, 128, 5), (33, 70, 96, 6), (90, 35, 64, 36), (150, 35, 64, 5)])
def test_layered_gru_policy_act_matches_oracle(rows, Do, Hd, K):
    """cm_gru_policy_act_ws: hidden state, sampled action and log-prob vs the CPU oracle + sampler; greedy (eps < 0) = argmax of the
    masked logits.  The last shape is a FUSED one: the workspace entry point must hand it to the fused step kernel unchanged."""
    from oracle import restatement as R
    from oracle import sampling
    from cleanmarl_amd import _native as N
    from cleanmarl_amd.learner import NetSpec, flatten_params, init_params_like_torch
    lib, dev = N.load(), torch.device("cuda:0")
    torch.manual_seed(9)
    spec = NetSpec(Do, Hd, 0, K, "gru")
    p = init_params_like_torch(spec)
    x = torch.randn(rows, Do); h0 = torch.randn(rows, Hd) * 0.5
    avail = torch.rand(rows, K) < 0.6
    avail[:, 1] = True
    d_x, d_av, d_p = x.to(dev), avail.to(torch.uint8).to(dev), flatten_params(p, dev)
    ws = torch.empty(max(16, lib.cm_gru_policy_act_workspace_bytes(rows, Do, Hd, K)), dtype=torch.uint8, device=dev)
    logits, h1 = R.gru_actor_logits(p, x, h0, avail)
    for eps in (0.0, -1.0):
        d_h = h0.clone().to(dev)
        act = torch.empty(rows, dtype=torch.int32, device=dev); lp = torch.empty(rows, device=dev)
        N.check(lib.cm_gru_policy_act_ws(N.ptr(d_x), Do, N.ptr(d_av), K, rows, Do, Hd, K, N.ptr(d_p), N.ptr(d_h), eps, 5, 77, 3,
                                         N.ptr(act), N.ptr(lp), 1, N.ptr(ws), ws.numel(), N.stream_ptr()), "gru act")
        assert _err(d_h.cpu().numpy(), h1.numpy()) <= TOL
        if eps == 0.0:
            a_ref, lp_ref, _ = sampling.act(logits.numpy(), avail.numpy(), 5, 77, 3)
            same = act.cpu().numpy() == a_ref
            assert same.mean() >= 0.99
            assert np.abs(lp.cpu().numpy()[same] - lp_ref[same]).max() <= TOL
        else:
            lg = logits.numpy()
            a_ref = lg.argmax(-1)
            same = act.cpu().numpy() == a_ref
            assert same.mean() >= 0.99  # ties / 1-ulp differences of nearly equal logits
            lse = torch.logsumexp(logits, -1).numpy()
            assert np.abs(lp.cpu().numpy()[same] - (lg.max(-1) - lse)[same]).max() <= TOL


def test_more_than_64_actions_is_an_explicit_error():
    """Heads of 33 .. 64 actions run on the layered schedule; beyond that both actor families refuse with a message (never a silent fallback)."""
    from cleanmarl_amd import _native as N
    lib, dev = N.load(), torch.device("cuda:0")
    z = torch.zeros(1 << 16, device=dev)
    ws = torch.empty(1 << 22, dtype=torch.uint8, device=dev)
    act = torch.zeros(8, dtype=torch.int32, device=dev)
    rc = lib.cm_policy_act_ws(N.ptr(z), 40, None, 65, 4, 40, 64, 1, 65, N.ptr(z), 0.0, 1, 0, 0, N.ptr(act), N.ptr(z), 1, N.ptr(ws), ws.numel(),
                              N.stream_ptr())
    assert rc != 0 and b"65" in lib.cm_last_error() and b"64" in lib.cm_last_error()
    rc = lib.cm_gru_policy_act_ws(N.ptr(z), 40, None, 65, 4, 40, 64, 65, N.ptr(z), N.ptr(z), 0.0, 1, 0, 0, N.ptr(act), N.ptr(z), 1, N.ptr(ws),
                                  ws.numel(), N.stream_ptr())
    assert rc != 0 and b"65" in lib.cm_last_error()


def test_fused_entry_points_still_refuse_layered_shapes_without_a_workspace():
    from cleanmarl_amd import _native as N
    lib, dev = N.load(), torch.device("cuda:0")
    z = torch.zeros(64, device=dev)
    rc = lib.cm_gru_policy_act(N.ptr(z), 70, None, 5, 1, 70, 64, 5, N.ptr(z), N.ptr(z), 1, 0, 0, None, None, 1, N.stream_ptr())
    assert rc != 0 and b"layered" in lib.cm_last_error()


@pytest.mark.parametrize("script,extra", [("ippo_lstm_multienvs", ["--synthetic_obs=105", "--actor_hidden_dim=64"]),
                                          ("mappo_lstm_multienvs", ["--synthetic_obs=40", "--actor_hidden_dim=128"])])
def test_recurrent_scripts_run_on_wide_shapes(script, extra, tmp_path, monkeypatch):
    """`*_lstm_multienvs.py` on a SMAClite-shaped observation (105 + 4 ids) and on a 128-wide GRU -- both were explicit errors: device
    rollout (per-step layered act), TBPTT update, greedy evaluation, all logged scalars finite."""
    from cleanmarl_amd.driver import run
    monkeypatch.chdir(tmp_path)
    out = run(script, ["--env_type=synthetic_shape", "--batch_size=6", "--synthetic_agents=4", "--synthetic_steps=12", "--synthetic_state=50",
                       "--synthetic_actions=17", "--total_timesteps=144", "--eval_steps=1", "--num_eval_ep=1", "--log_every=1",
                       "--critic_hidden_dim=64", "--tbptt=5", "--greedy_eval"] + extra)
    assert out["training_step"] >= 3 and all(math.isfinite(v) for _, v, _ in out["history"])
    assert "eval/ep_reward" in {t for t, _, _ in out["history"]}


def test_config5_full_size_chunk_pass_is_additive_over_env_shards_and_deterministic():
    """BASELINE config 5 at its full size (1024 envs x 5 agents x 128 steps, GRU hidden 64, tbptt 10), where no CPU oracle finishes in
    seconds: the TBPTT chunk pass (cm_gru_actor_chunk_fwd_bwd, second-generation 32-row sweeps) is a SUM over sequences, so the
    un-normalised gradient + statistics buffer of the full batch equals the sum over uneven, tile-unaligned env shards (the property the
    env-sharded run relies on), h_out rows are those of the shards, two runs give identical bits, everything is finite -- and one shard
    (24 envs) is small enough for the oracle: its chunk gradient matches the CPU restatement, which pins the full-size pass through the
    additivity."""
    from oracle import restatement as R
    from cleanmarl_amd import _native as N
    from cleanmarl_amd.learner import DeviceBatch, NetSpec, flatten_params, init_params_like_torch
    lib, dev = N.load(), torch.device("cuda:0")
    E, A, T, K, H, tb = 1024, 5, 128, 5, 64, 10
    Do, Ds = 7 * A, 6 * A * A
    g = torch.Generator().manual_seed(0)
    b = DeviceBatch(E, A, T, Do, Ds, K, dev)
    b.obs.copy_(torch.randn(E, A, T, Do, generator=g)); b.avail.fill_(1)
    b.action.copy_(torch.randint(0, K, (E, A, T), generator=g).int()); b.logp.copy_(-1.6 + 0.1 * torch.randn(E, A, T, generator=g))
    b.adv.copy_(torch.randn(E, A, T, generator=g)); b.ep_len.copy_(torch.randint(T // 2, T + 1, (E,), generator=g).int())
    torch.manual_seed(1)
    spec = NetSpec(Do, H, 0, K, "gru")
    plist = init_params_like_torch(spec)
    p = flatten_params(plist, dev)
    P = p.numel()
    ws = torch.empty(lib.cm_gru_workspace_bytes(E, A, Do, H, K, tb), dtype=torch.uint8, device=dev)
    t0, t1 = 60, 70  # a chunk that straddles the shortest episodes' ends
    h_in = torch.randn(E * A, H, generator=g).to(dev) * 0.3

    def chunk(lo, hi):
        n = hi - lo
        gbuf = torch.zeros(P + N.NUM_STATS, device=dev)
        h_out = torch.zeros(n * A, H, device=dev)
        N.check(lib.cm_gru_actor_chunk_fwd_bwd(N.ptr(b.obs[lo:hi]), N.ptr(b.avail[lo:hi]), N.ptr(b.action[lo:hi]), N.ptr(b.logp[lo:hi]),
                                               N.ptr(b.adv[lo:hi]), N.ptr(b.ep_len[lo:hi]), n, A, T, t0, t1, Do, H, K, N.ptr(p),
                                               N.ptr(h_in[lo * A:hi * A]), N.ptr(h_out), 0.2, 1e-3, N.ptr(gbuf), N.ptr(ws), ws.numel(),
                                               N.stream_ptr()), "cm_gru_actor_chunk_fwd_bwd")
        torch.cuda.synchronize()
        return gbuf, h_out
    full, h_full = chunk(0, E)
    again, _ = chunk(0, E)
    assert torch.equal(full, again) and torch.isfinite(full).all() and torch.isfinite(h_full).all()
    parts, first = torch.zeros_like(full), None
    for lo, hi in ((0, 24), (24, 333), (333, E)):
        gp, hp_ = chunk(lo, hi)
        parts += gp
        assert (hp_ - h_full[lo * A:hi * A]).abs().max().item() <= 1e-5
        if first is None:
            first = gp.clone()
    assert (full - parts).abs().max().item() <= 1e-4 * full.abs().max().item()
    valid = ((torch.arange(t0, t1)[None, :] < b.ep_len.cpu()[:, None]).sum()).item()
    assert full[P + N.STAT_COUNT].item() == valid
    # ---- the 24-env shard against the oracle's chunk gradient
    sb_len = b.ep_len[:24].cpu()
    mask = torch.arange(T)[None, :] < sb_len[:, None]
    batch = dict(obs=b.obs[:24].permute(0, 2, 1, 3).cpu(), actions=b.action[:24].permute(0, 2, 1).long().cpu(),
                 log_probs=b.logp[:24].permute(0, 2, 1).cpu(), avail=b.avail[:24].permute(0, 2, 1, 3).bool().cpu(), mask=mask)
    og, ostats = R.gru_chunk_sums(plist, batch, b.adv[:24].permute(0, 2, 1).cpu(), h_in[:24 * A].cpu(), t0, t1, 0.2, 1e-3)
    assert _err(first[:P].cpu().numpy(), R.flat(og).numpy()) <= TOL * max(1.0, float(first[:P].abs().max()))
    assert abs(float(first[P + N.STAT_COUNT]) - ostats["count"]) == 0
