"""Pins that do not rest on the build's word alone ("self-referential pins"):

* the counter RNG -- oracle/philox.py (plain-int restatement of the published algorithm), cleanmarl_amd/env/philox.py (numpy twin),
  cm_philox4x32 on the host and on the device -- against the Random123 Philox4x32-10 known-answer vectors;
* the action sampler against the DISTRIBUTION the reference samples from (Categorical(logits=masked logits),
  cleanmarl/mappo_multienvs.py:172-183): chi-square of cm_policy_act's actions at K in {5, 17, 36} with availability masks;
* the host collation (RolloutBuffer.add / get_batch, cleanmarl/mappo_multienvs.py:82-157, fed by the rollout loop :393-453) against the
  batches the UNMODIFIED reference collated for the ragged goldens, by replaying the goldens' actions on the env they were captured on
  (tests/stub_envs.py::SynthEnv) through every vector env of the build;
* the CLI defaults against the ``hp_*`` arrays the goldens hold (the reference's own ``Args`` as it ran).
"""
import importlib.util
import os

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")


def _kat_words():
    from oracle.philox import KAT
    ck = np.array([list(c) + list(k) for c, k, _ in KAT], dtype=np.uint32)
    want = np.array([list(e) for _, _, e in KAT], dtype=np.uint32)
    return ck, want


def _random_words(n=4096, seed=7):
    rng = np.random.default_rng(seed)
    return rng.integers(0, 2 ** 32, size=(n, 6), dtype=np.uint64).astype(np.uint32)


# ------------------------------------------------------------------------------------------------------------------------ Philox
def test_philox_restatement_and_numpy_twin_match_random123_known_answers():
    from cleanmarl_amd.env.philox import philox4x32
    from oracle.philox import KAT, philox4x32_10
    for ctr, key, want in KAT:
        assert philox4x32_10(ctr, key) == want
        got = philox4x32(*[np.uint32(c) for c in ctr], key[0], key[1])
        assert tuple(int(x) for x in got) == want
    ck = _random_words(512)
    twin = np.stack(philox4x32(ck[:, 0], ck[:, 1], ck[:, 2], ck[:, 3], 0, 0), 1)  # the twin takes scalar keys: check per key below
    for i in range(0, 512, 37):
        one = philox4x32(ck[i, 0], ck[i, 1], ck[i, 2], ck[i, 3], int(ck[i, 4]), int(ck[i, 5]))
        assert tuple(int(x) for x in one) == philox4x32_10(ck[i, :4], ck[i, 4:])
    assert tuple(int(x) for x in twin[5]) == philox4x32_10(ck[5, :4], (0, 0))


def test_library_host_philox_matches_known_answers_and_the_restatement():
    """cm_philox4x32 (csrc/cm_common.h, __host__ __device__: the same source the kernels compile) through cm_philox4x32_host."""
    from cleanmarl_amd import _native as N
    from cleanmarl_amd.build import build_native
    from oracle.philox import philox4x32_10
    build_native()
    lib = N.load()
    ck, want = _kat_words()
    out = np.zeros((len(ck), 4), np.uint32)
    assert lib.cm_philox4x32_host(ck.ctypes.data, len(ck), out.ctypes.data) == 0
    assert np.array_equal(out, want)
    ck = _random_words(2048)
    out = np.zeros((len(ck), 4), np.uint32)
    assert lib.cm_philox4x32_host(ck.ctypes.data, len(ck), out.ctypes.data) == 0
    for i in range(0, 2048, 13):
        assert tuple(int(x) for x in out[i]) == philox4x32_10(ck[i, :4], ck[i, 4:])


@pytest.mark.gpu
def test_device_philox_matches_known_answers_and_the_host_function():
    from cleanmarl_amd import _native as N
    lib = N.load()
    dev = torch.device("cuda:0")
    kat, want = _kat_words()
    ck = np.concatenate([kat, _random_words(100000)])
    d_in = torch.from_numpy(ck.view(np.int32)).to(dev)
    d_out = torch.zeros(len(ck), 4, dtype=torch.int32, device=dev)
    N.check(lib.cm_philox4x32_device(N.ptr(d_in), len(ck), N.ptr(d_out), N.stream_ptr()), "cm_philox4x32_device")
    got = d_out.cpu().numpy().view(np.uint32)
    assert np.array_equal(got[:len(kat)], want)
    host = np.zeros((len(ck), 4), np.uint32)
    assert lib.cm_philox4x32_host(ck.ctypes.data, len(ck), host.ctypes.data) == 0
    assert np.array_equal(got, host)


@pytest.mark.gpu
def test_act_kernel_draws_the_restatements_uniforms():
    """The sampler's key layout (seed -> key words, (row, t, stream) -> counter words) as oracle/philox.py::act_uniforms states it:
    with two equally likely actions the action IS the uniform's comparison with 1/2."""
    from cleanmarl_amd import _native as N
    from cleanmarl_amd.learner import NetSpec, flatten_params, init_params_like_torch
    from oracle.philox import act_uniforms
    lib, dev = N.load(), torch.device("cuda:0")
    rows, Do, K = 3000, 8, 2
    spec = NetSpec(Do, 64, 1, K)
    p = [torch.zeros_like(q) for q in init_params_like_torch(spec)]  # all-zero network: logits 0, 0 -> p = 1/2 each
    x = torch.zeros(rows, Do, device=dev); av = torch.ones(rows, K, dtype=torch.uint8, device=dev)
    act = torch.empty(rows, dtype=torch.int32, device=dev); lp = torch.empty(rows, device=dev)
    seed, row_offset, t = 0x1234567887654321, (1 << 33) + 5, 17
    N.check(lib.cm_policy_act(N.ptr(x), Do, N.ptr(av), K, rows, Do, 64, 1, K, N.ptr(flatten_params(p, dev)), seed, row_offset, t,
                              N.ptr(act), N.ptr(lp), 1, N.stream_ptr()), "cm_policy_act")
    u = act_uniforms(rows, seed, row_offset, t)
    assert np.array_equal(act.cpu().numpy(), (u >= 0.5).astype(np.int32))  # inverse CDF: action 0 iff u * 2 < 1
    assert np.allclose(lp.cpu().numpy(), np.log(0.5), atol=1e-6)


# ------------------------------------------------------------------------------------------------------- sampler vs Categorical
@pytest.mark.gpu
@pytest.mark.parametrize("K,H,L", [(5, 64, 1), (17, 64, 1), (36, 64, 1)])
def test_sampled_actions_follow_the_reference_categorical(K, H, L):
    """chi-square goodness of fit of cm_policy_act_ws's actions against softmax(masked logits) -- the distribution
    Categorical(logits=...) of cleanmarl/mappo_multienvs.py:172-183 defines -- per (observation, mask) group; probabilities from the
    oracle's logits.  K = 5 (fused quad sampler), 17 (fused, K > 8), 36 (layered schedule's 64-wide head)."""
    from scipy import stats
    from cleanmarl_amd import _native as N
    from cleanmarl_amd.learner import NetSpec, flatten_params, init_params_like_torch
    from oracle import restatement as R
    lib, dev = N.load(), torch.device("cuda:0")
    G, M, Do = 12, 40000, 23
    spec = NetSpec(Do, H, L, K)
    torch.manual_seed(K)
    p = [q * 2.5 for q in init_params_like_torch(spec)]  # peaked enough that the categories differ by orders of magnitude
    g = torch.Generator().manual_seed(100 + K)
    xg = torch.randn(G, Do, generator=g)
    avg = torch.rand(G, K, generator=g) < 0.6
    avg[:, 0] = True
    avg[0] = True              # one group with every action available
    avg[1] = False; avg[1, 3] = True  # one with a single legal action
    x = xg.repeat_interleave(M, 0).contiguous(); av = avg.repeat_interleave(M, 0).contiguous()
    rows = G * M
    d_x, d_av = x.to(dev), av.to(torch.uint8).to(dev)
    act = torch.empty(rows, dtype=torch.int32, device=dev); lp = torch.empty(rows, device=dev)
    need = lib.cm_policy_act_workspace_bytes(rows, Do, H, L, K)
    ws = torch.empty(max(need, 16), dtype=torch.uint8, device=dev)
    N.check(lib.cm_policy_act_ws(N.ptr(d_x), Do, N.ptr(d_av), K, rows, Do, H, L, K, N.ptr(flatten_params(p, dev)), 0.0, 99, 12345, 3,
                                 N.ptr(act), N.ptr(lp), 1, N.ptr(ws) if need else None, need, N.stream_ptr()), "cm_policy_act_ws")
    a = act.cpu().numpy().reshape(G, M); lpn = lp.cpu().numpy().reshape(G, M)
    logits = R.actor_logits(p, xg, avg).double()
    logp = torch.log_softmax(logits, -1).numpy()
    pvals = []
    for gi in range(G):
        legal = avg[gi].numpy()
        assert legal[a[gi]].all()  # never an unavailable action
        assert np.abs(lpn[gi] - logp[gi][a[gi]]).max() <= 1e-4  # the log-prob that goes into the rollout buffer
        prob = np.exp(logp[gi])[legal]
        cnt = np.bincount(a[gi], minlength=K)[legal].astype(np.float64)
        keep = prob * M >= 5.0  # chi-square validity: pool the rare categories
        obs_c = np.append(cnt[keep], cnt[~keep].sum()); exp_c = np.append(prob[keep], prob[~keep].sum()) * M
        if exp_c[-1] < 5.0:  # still too rare: fold into the smallest kept cell
            j = np.argmin(exp_c[:-1]) if len(exp_c) > 1 else 0
            obs_c[j] += obs_c[-1]; exp_c[j] += exp_c[-1]; obs_c, exp_c = obs_c[:-1], exp_c[:-1]
        if len(exp_c) < 2:
            assert cnt.sum() == M
            continue
        chi2 = ((obs_c - exp_c) ** 2 / exp_c).sum()
        pvals.append(stats.chi2.sf(chi2, len(exp_c) - 1))
    assert len(pvals) >= G - 2
    assert min(pvals) > 1e-5, pvals        # no group is off (a wrong CDF direction / mask / off-by-one fails at p ~ 0)
    assert stats.combine_pvalues(pvals, method="fisher")[1] > 1e-4


# ---------------------------------------------------------------------------------------------------- host collation vs goldens
def _cases():
    spec = importlib.util.spec_from_file_location("make_golden", os.path.join(GOLDEN, "make_golden.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)  # importing it never touches /root/reference (only running it does)
    return mod.CASES


class _Replay:
    """Stands in for the actor: hands out the golden's actions / log-probs of step t for the envs still alive (the reference's alive
    list shrinks in env order, cleanmarl/mappo_multienvs.py:437-452)."""

    def __init__(self, z):
        z = _golden_in_env_order(z)
        self.a, self.lp, self.mask, self.t = z["b_actions"], z["b_log_probs"], z["b_mask"], 0

    def act_rows(self):
        alive = np.flatnonzero(self.mask[:, self.t])
        a, l = self.a[alive, self.t], self.lp[alive, self.t]
        self.t += 1
        return alive, a, l

    def act(self, obs, avail, h=None, seed=0, eps=0.0, **kw):
        alive, a, l = self.act_rows()
        assert obs.shape[0] == len(alive)
        return a.reshape(-1).astype(np.int32), l.reshape(-1).astype(np.float32), None


def _patch_factory(monkeypatch, name):
    """Route the build's env factory to the goldens' env: env i of a vector env is SynthEnv(env_id = i)."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import stub_envs
    from cleanmarl_amd.env import shm_vector, vector
    script, ov, env_spec = _cases()[name]
    monkeypatch.setattr(stub_envs.SynthEnv, "spec", dict(stub_envs.SynthEnv.spec, **env_spec))
    fac = lambda env_type=None, env_name=None, env_family=None, agent_ids=True, kwargs=None, index=0, seed=1, synthetic=None: \
        stub_envs.SynthEnv(agent_ids, env_id=index)
    monkeypatch.setattr(vector, "environment", fac)
    monkeypatch.setattr(shm_vector, "environment", fac)  # forked workers inherit the patched modules
    return ov["batch_size"]


def _assert_is_golden_batch(b, z, stats):
    """DeviceBatch [E, A, T, F] == the reference's collated batch [B, T, A, F] (b_reward_raw: before the reward normalisation, which the
    learner applies -- a7 / test_update_matches_reference_golden).  ORDER: the reference appends an episode to its RolloutBuffer when it
    ENDS (cleanmarl/mappo_multienvs.py:437-452), so its batch is in completion order -- shorter episodes first, ties in env order; the
    build keeps env order (a contiguous env range per rank is what the sharding relies on).  Every loss term is a sum over the batch, so
    the permutation changes nothing but the association of fp32 sums; it is made explicit here: golden row i = build row perm[i]."""
    raw = z["b_reward_raw"] if "b_reward_raw" in z.files else z["b_reward"]
    ep_len = b.ep_len.cpu().numpy()
    perm = torch.from_numpy(np.argsort(ep_len, kind="stable"))
    assert len(set(ep_len.tolist())) > 1 and not np.array_equal(perm.numpy(), np.arange(len(ep_len)))  # ragged, and the order does differ
    assert torch.equal(b.obs.permute(0, 2, 1, 3).cpu()[perm], torch.from_numpy(z["b_obs"]))
    assert torch.equal(b.state.cpu()[perm], torch.from_numpy(z["b_states"]))
    assert torch.equal(b.avail.permute(0, 2, 1, 3).cpu().bool()[perm], torch.from_numpy(z["b_avail_actions"]).bool())
    assert torch.equal(b.action.permute(0, 2, 1).cpu().long()[perm], torch.from_numpy(z["b_actions"]).long())
    assert torch.equal(b.logp.permute(0, 2, 1).cpu()[perm], torch.from_numpy(z["b_log_probs"]))
    assert torch.equal(b.reward.cpu()[perm], torch.from_numpy(raw))
    mask = z["b_mask"]
    assert ep_len[perm.numpy()].tolist() == mask.sum(1).tolist() and list(stats["ep_len"]) == ep_len.tolist()
    np.testing.assert_allclose(np.asarray(stats["ep_reward"])[perm.numpy()], (raw.astype(np.float64) * mask).sum(1), rtol=1e-6, atol=1e-6)


def _golden_in_env_order(z):
    """The golden's arrays re-ordered from completion order to env order (what the replayed actor must hand out): env e ends after
    horizon(e) steps, the reference's row of env e is its rank in the stable sort by length."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import stub_envs
    B = z["b_obs"].shape[0]
    lens = np.array([stub_envs.SynthEnv(True, env_id=e).horizon for e in range(B)])
    perm = np.argsort(lens, kind="stable")     # golden row i = env perm[i]
    inv = np.argsort(perm)                     # env e = golden row inv[e]
    assert z["b_mask"].sum(1).tolist() == lens[perm].tolist()
    return {k: z[k][inv] for k in ("b_actions", "b_log_probs", "b_mask")}


@pytest.mark.parametrize("venv_kind", ["pipe", "shm"])
@pytest.mark.parametrize("name", ["mappo_ragged_norm", "ippo_ragged_norm"])
def test_host_collation_reproduces_the_reference_batch(name, venv_kind, monkeypatch):
    from cleanmarl_amd.driver import host_rollout, host_rollout_shm
    from cleanmarl_amd.env.shm_vector import ShmVectorEnv
    from cleanmarl_amd.env.vector import PipeVectorEnv
    E = _patch_factory(monkeypatch, name)
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    A = z["b_obs"].shape[2]
    fac = dict(env_type="pz", env_name="x", env_family="mpe", agent_ids=True, kwargs={}, seed=1, synthetic=None)
    if venv_kind == "pipe":
        venv = PipeVectorEnv(E, fac)
        b, stats = host_rollout(venv, _Replay(z), E, A, 0, False, torch.device("cpu"))
    else:
        venv = ShmVectorEnv(E, fac, n_workers=3)
        b, stats = host_rollout_shm(venv, _Replay(z), E, A, 0, False, torch.device("cpu"))
    venv.close()
    _assert_is_golden_batch(b, z, stats)


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["mappo_ragged_norm", "ippo_ragged_norm"])
def test_pinned_host_rollout_reproduces_the_reference_batch(name, monkeypatch):
    """The default host path (page-locked shared blocks -> device rollout buffer, no host collation) with the act step replaced by the
    golden's actions: what lands in the learner's DeviceBatch is the reference's batch."""
    from cleanmarl_amd.env.shm_vector import ShmVectorEnv
    from cleanmarl_amd.host_rollout import PinnedHostRollout
    from cleanmarl_amd.learner import HParams, NetSpec, PPOLearner
    E = _patch_factory(monkeypatch, name)
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    B, T, A, Do = z["b_obs"].shape
    K, Ds = z["b_avail_actions"].shape[-1], z["b_states"].shape[-1]
    dev = torch.device("cuda:0")
    fac = dict(env_type="pz", env_name="x", env_family="mpe", agent_ids=True, kwargs={}, seed=1, synthetic=None)
    L = PPOLearner("mappo", NetSpec(Do, 64, 1, K), NetSpec(Ds, 64, 1, 1), A, HParams(), dev)
    zo = _golden_in_env_order(z)
    acts = torch.from_numpy(zo["b_actions"]).to(dev).int(); lps = torch.from_numpy(zo["b_log_probs"]).to(dev)
    for pad in (False, True):
        venv = ShmVectorEnv(E, fac, n_workers=2)  # fresh envs per pass: SynthEnv's episode counter is part of its transition function
        ph = PinnedHostRollout(venv, L, False, dev, t_cap=4, pad=pad)  # t_cap 4: the buffer grows twice on the way

        def replay(t, seed, eps, ph=ph):
            ph.buf.action[:, :, t] = acts[:, t]
            ph.buf.logp[:, :, t] = lps[:, t]
        monkeypatch.setattr(ph, "_act", replay)
        b, stats = ph.collect(0)
        torch.cuda.synchronize()
        _assert_is_golden_batch(b, z, stats)
        ph.close()
        venv.close()


# ---------------------------------------------------------------------------------------------------------------- CLI defaults
_SCRIPT_OF = {"mappo_dense": "mappo_multienvs", "ippo_dense": "ippo_multienvs", "mappo_lstm_dense": "mappo_lstm_multienvs",
              "ippo_lstm_ragged": "ippo_lstm_multienvs", "coma_default_width": "coma_multienvs"}


@pytest.mark.parametrize("name,script", sorted(_SCRIPT_OF.items()))
def test_cli_defaults_are_the_reference_args_as_the_goldens_recorded_them(name, script):
    """Every golden stores the reference's complete ``Args`` (dataclass fields of the script that produced it) as hp_*; all fields the
    capture did not override are the reference's defaults -- the build's parser must agree on each of them (and know every field)."""
    from cleanmarl_amd.args import parse_args
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    _, overrides, _ = _cases()[name]
    forced = set(overrides) | {"env_type", "total_timesteps", "eval_steps", "seed"}  # run_reference's own overrides
    a = parse_args(script, [])
    checked = 0
    for key in z.files:
        if not key.startswith("hp_"):
            continue
        f = key[3:]
        if f in forced or f == "device":  # device: this build is GPU-only on purpose
            continue
        assert hasattr(a, f), f"{script}: the reference's --{f} is unknown to the build's parser"
        ref = z[key]
        ref = ref.item() if ref.dtype.kind in "fiub" else str(ref)
        got = getattr(a, f)
        if isinstance(ref, float) and not isinstance(got, str):
            assert float(got) == ref, (script, f, got, ref)
        else:
            assert str(got) == str(ref), (script, f, got, ref)
        checked += 1
    assert checked >= 20, checked
    for f in forced - {"env_type"}:
        assert hasattr(a, f)


# ------------------------------------------------------------------------------------------ roofline.traffic is stamped, never stale
def test_bench_refuses_a_pmc_record_taken_on_other_sources(tmp_path):
    """bench.py's roofline.traffic is read from a committed PMC record (the counters cannot be collected inside the timed process): the
    record must carry the hash of the sources of the library being timed and name the actor pass; anything else leaves traffic null."""
    import json
    import sys
    sys.path.insert(0, ROOT)
    import bench
    from cleanmarl_amd.build import source_hash
    good = dict(kernel="void (anonymous namespace)::k_mlp<1, 2, 1, 1, 2, false, true>((anonymous namespace)::MlpArgs)",
                hbm_bytes_per_launch=1.03e9, mfma_busy_frac=0.66, source="x", workload="cfg3", source_hash=source_hash())
    p = tmp_path / "pmc.json"
    p.write_text(json.dumps(good))
    rec, why = bench.load_pmc(str(p))
    assert why is None and rec["hbm_bytes_per_launch"] == 1.03e9
    for patch, word in ((dict(source_hash="0123456789abcdef"), "stale"), (dict(source_hash=None), "stale"),
                        (dict(kernel="k_critic_fused<6>"), "kernel"), (dict(workload="cfg4"), "workload")):
        p.write_text(json.dumps(dict(good, **patch)))
        rec, why = bench.load_pmc(str(p))
        assert rec is None and word in why, (patch, why)
    p.write_text("{not json")
    assert bench.load_pmc(str(p))[0] is None
    assert bench.load_pmc(str(tmp_path / "missing.json"))[0] is None
    # the committed record is either current or refused -- never silently used for other kernels
    rec, why = bench.load_pmc(os.path.join(ROOT, "profiles", "pmc_dominant_kernel.json"))
    assert (rec is None) != (why is None)


def test_bench_refuses_issue_counters_taken_on_other_sources(tmp_path):
    """roofline.issue (MFMA-busy and VALU-issue shares of the dominant kernel's launch) comes from a committed PMC record as well: same
    stamp rule as roofline.traffic -- other sources, another kernel or an unreadable file give None, never numbers of another build."""
    import json
    import sys
    sys.path.insert(0, ROOT)
    import bench
    from cleanmarl_amd.build import source_hash
    name = "void (anonymous namespace)::k_mlp<1, 2, 1, 1, 2, false, true>((anonymous namespace)::MlpArgs)"
    ctr = dict(SQ_INSTS_MFMA=50.0e6, SQ_INSTS_VALU=170.0e6, SQ_VALU_MFMA_BUSY_CYCLES=2.9e9, GRBM_GUI_ACTIVE=32.0e6, SQ_VALU_MFMA_COEXEC_CYCLES=0.0)
    p = tmp_path / "issue.json"
    p.write_text(json.dumps(dict(source_hash=source_hash(), workload="cfg3", kernels={name: ctr})))
    rec = bench.load_issue_counters(str(p), bench.DOMINANT_KERNEL)
    simd = 1024 * 32.0e6 / 8
    assert rec["other_valu_insts_per_launch"] == 120.0e6 and abs(rec["mfma_busy_frac"] - 2.9e9 / simd) < 1e-12
    assert abs(rec["valu_issue_frac_at_4_cycles"] - 4 * 120.0e6 / simd) < 1e-12 and rec["valu_mfma_coexec_cycles"] == 0.0
    p.write_text(json.dumps(dict(source_hash="0123456789abcdef", workload="cfg3", kernels={name: ctr})))
    assert bench.load_issue_counters(str(p), bench.DOMINANT_KERNEL) is None
    p.write_text(json.dumps(dict(source_hash=source_hash(), workload="cfg3", kernels={"k_critic_fused<6>": ctr})))
    assert bench.load_issue_counters(str(p), bench.DOMINANT_KERNEL) is None
    p.write_text("{not json")
    assert bench.load_issue_counters(str(p), bench.DOMINANT_KERNEL) is None
    assert bench.load_issue_counters(str(tmp_path / "missing.json"), bench.DOMINANT_KERNEL) is None
