"""End-to-end runs of the four drop-in scripts on the GPU (tiny budgets): reference plumbing of config 1
(pipe-protocol envs with the bundled CPU MPE-like env) and the on-device synthetic env."""
import pytest

pytestmark = pytest.mark.gpu

TAGS = {"train/critic_loss", "train/actor_loss", "train/entropy", "train/kl_divergence", "train/clipped_ratios",
        "train/actor_gradients", "train/critic_gradients", "train/num_updates"}


@pytest.mark.parametrize("script", ["mappo_multienvs", "ippo_multienvs", "mappo_lstm_multienvs", "ippo_lstm_multienvs"])
@pytest.mark.parametrize("env_type", ["synthetic", "synthetic_cpu"])
def test_script_runs_and_logs_reference_tags(script, env_type, tmp_path, monkeypatch):
    import math
    from cleanmarl_amd.driver import run
    monkeypatch.chdir(tmp_path)
    out = run(script, [f"--env_type={env_type}", "--env_name=simple_spread_v3", "--batch_size=4", "--synthetic_agents=3",
                       "--synthetic_steps=25", "--total_timesteps=250", "--eval_steps=1", "--num_eval_ep=2", "--log_every=1",
                       "--tbptt=10"])
    tags = {t for t, _, _ in out["history"]}
    assert TAGS <= tags and {"rollout/ep_reward", "rollout/ep_length", "rollout/num_episodes"} <= tags
    assert {"eval/ep_reward", "eval/std_ep_reward", "eval/ep_length"} <= tags
    assert all(math.isfinite(v) for _, v, _ in out["history"])
    assert out["step"] >= 250 and out["training_step"] == 3 * math.ceil(250 / 100)
    steps = [s for t, _, s in out["history"] if t == "train/num_updates"]
    assert steps == sorted(steps) and steps[0] == 100  # x-axis = env steps (4 envs x 25 steps), like the reference


@pytest.mark.parametrize("script", ["mappo", "ippo", "mappo_lstm", "ippo_lstm"])
def test_single_env_front_ends(script, tmp_path, monkeypatch):
    """cleanmarl/mappo.py, ippo.py, mappo_lstm.py, ippo_lstm.py: batch_size episodes collected sequentially from ONE
    in-process env (ragged episode lengths via the CPU env's per-index horizon are not needed: fixed 25 steps)."""
    import math
    from cleanmarl_amd.driver import run
    monkeypatch.chdir(tmp_path)
    out = run(script, ["--env_type=synthetic_cpu", "--batch_size=3", "--synthetic_agents=3", "--synthetic_steps=25",
                       "--total_timesteps=150", "--eval_steps=1", "--num_eval_ep=1", "--log_every=1"])
    tags = {t for t, _, _ in out["history"]}
    assert TAGS <= tags and {"rollout/ep_reward", "rollout/ep_length", "rollout/num_episodes", "eval/ep_reward"} <= tags
    assert all(math.isfinite(v) for _, v, _ in out["history"])
    assert out["step"] == 150 and out["training_step"] == 6
    ne = [v for t, v, _ in out["history"] if t == "rollout/num_episodes"]
    assert ne and ne[-1] == 6  # 2 iterations x batch_size 3


@pytest.mark.parametrize("script", ["ippo_multienvs", "ippo_lstm_multienvs"])
def test_shape_env_scripts(script, tmp_path, monkeypatch):
    """IPPO on the SMAClite-shaped device env (availability masks, 17 actions, obs 33 + ids) and on its CPU twin."""
    import math
    from cleanmarl_amd.driver import run
    monkeypatch.chdir(tmp_path)
    for env_type in ("synthetic_shape", "synthetic_shape_cpu"):
        out = run(script, [f"--env_type={env_type}", "--batch_size=6", "--synthetic_agents=4", "--synthetic_steps=12",
                           "--synthetic_obs=33", "--synthetic_state=50", "--synthetic_actions=17", "--total_timesteps=144",
                           "--eval_steps=1", "--num_eval_ep=1", "--log_every=1", "--critic_hidden_dim=64"])
        assert TAGS <= {t for t, _, _ in out["history"]} and all(math.isfinite(v) for _, v, _ in out["history"])


@pytest.mark.parametrize("mfma", ["", "bf16"])
def test_learning_signal_on_synthetic_env(mfma, tmp_path, monkeypatch):
    """A few hundred iterations of MAPPO on the on-device env must improve the episode return -- with the default exact-fp32 arithmetic
    and with the opt-in single-pass bf16 training passes (CM_MFMA=bf16, looser parity tier: it must still learn)."""
    from cleanmarl_amd.driver import run
    monkeypatch.chdir(tmp_path)
    if mfma:
        monkeypatch.setenv("CM_MFMA", mfma)
    out = run("mappo_multienvs", ["--env_type=synthetic", "--batch_size=256", "--synthetic_agents=3", "--synthetic_steps=25",
                                  "--total_timesteps=1280000", "--eval_steps=100000", "--log_every=1",
                                  "--actor_hidden_dim=64", "--normalize_advantage"])
    r = [v for t, v, _ in out["history"] if t == "rollout/ep_reward"]
    assert len(r) >= 100
    first, last = sum(r[:10]) / 10, sum(r[-10:]) / 10
    assert last > first + 0.05 * abs(first), (first, last)


def test_learning_signal_gru(tmp_path, monkeypatch):
    """The GRU / TBPTT scripts (fused GRU rollout + 32-row sweeps) must improve the episode return of the on-device env too."""
    from cleanmarl_amd.driver import run
    monkeypatch.chdir(tmp_path)
    common = ["--env_type=synthetic", "--batch_size=256", "--synthetic_agents=3", "--synthetic_steps=25", "--total_timesteps=960000",
              "--eval_steps=100000", "--log_every=1", "--actor_hidden_dim=64", "--tbptt=10"]
    for script in ("mappo_lstm_multienvs", "ippo_lstm_multienvs"):
        out = run(script, common)
        r = [v for t, v, _ in out["history"] if t == "rollout/ep_reward"]
        first, last = sum(r[:15]) / 15, sum(r[-15:]) / 15
        assert last > first + 0.05 * abs(first), (script, first, last)


def test_checkpoint_resume_is_bit_exact(tmp_path, monkeypatch):
    """train 4 iterations in one go == train 2, checkpoint, resume 2 (replicated state + counter-based env/action RNG)."""
    import torch
    from cleanmarl_amd.driver import run
    monkeypatch.chdir(tmp_path)
    common = ["--env_type=synthetic", "--batch_size=16", "--synthetic_agents=3", "--synthetic_steps=10", "--eval_steps=100000",
              "--greedy_eval"]
    full = run("mappo_multienvs", common + ["--total_timesteps=640"])
    ck = str(tmp_path / "ck.pt")
    run("mappo_multienvs", common + ["--total_timesteps=320", f"--checkpoint={ck}"])
    res = run("mappo_multienvs", common + ["--total_timesteps=640", f"--checkpoint={ck}"])
    assert res["training_step"] == full["training_step"] == 12
    assert torch.equal(res["learner"].actor, full["learner"].actor) and torch.equal(res["learner"].critic, full["learner"].critic)
    assert torch.equal(res["learner"].opt_a.v, full["learner"].opt_a.v)


@pytest.mark.parametrize("vector_env", ["pipe", "shm", "pinned"])
def test_config1_pz_simple_spread_end_to_end(vector_env, tmp_path, monkeypatch):
    """BASELINE.json configs[0] verbatim -- `mappo_multienvs.py --env_type=pz --env_family=mpe --env_name=simple_spread_v3`, 4 envs --
    through the PettingZoo adapter (stand-in package, tests/stub_envs.py: pettingzoo is not installable here), the vector env, the
    HIP actor / learner and the logger."""
    import math
    from stub_envs import install
    from cleanmarl_amd.driver import run
    install(monkeypatch)
    monkeypatch.chdir(tmp_path)
    out = run("mappo_multienvs", ["--env_type=pz", "--env_family=mpe", "--env_name=simple_spread_v3", "--batch_size=4",
                                  f"--vector_env={vector_env}", "--total_timesteps=250", "--eval_steps=1", "--num_eval_ep=2",
                                  "--log_every=1"])
    tags = {t for t, _, _ in out["history"]}
    assert TAGS <= tags and {"rollout/ep_reward", "rollout/ep_length", "eval/ep_reward"} <= tags
    assert all(math.isfinite(v) for _, v, _ in out["history"])
    assert out["step"] == 300 and out["training_step"] == 9  # 3 iterations x (4 envs x 25 steps), 3 epochs each
    assert [v for t, v, _ in out["history"] if t == "rollout/ep_length"][0] == 25.0


def test_smaclite_adapter_end_to_end(tmp_path, monkeypatch):
    """`ippo_multienvs.py --env_type=smaclite --env_name=3m` through the SMAClite adapter (stand-in package): availability masks
    from the env reach the sampler (the fake env asserts that no illegal action arrives), battle_won is logged."""
    import math
    from stub_envs import install
    from cleanmarl_amd.driver import run
    install(monkeypatch)
    monkeypatch.chdir(tmp_path)
    out = run("ippo_multienvs", ["--env_type=smaclite", "--env_name=3m", "--batch_size=3", "--total_timesteps=600", "--eval_steps=1",
                                 "--num_eval_ep=1", "--log_every=1"])
    tags = {t for t, _, _ in out["history"]}
    assert TAGS <= tags and {"rollout/battle_won", "eval/battle_won", "rollout/ep_length"} <= tags
    assert all(math.isfinite(v) for _, v, _ in out["history"])
    assert [v for t, v, _ in out["history"] if t == "rollout/ep_length"][0] == 150.0  # TimeLimit(150)


def _host_learner(A, Do, Ds, K, recurrent, dev):
    import torch
    from cleanmarl_amd.learner import HParams, NetSpec, PPOLearner
    from cleanmarl_amd.gru import GRUPPOLearner
    torch.manual_seed(4)
    aspec = NetSpec(Do, 64, 0 if recurrent else 1, K, "gru" if recurrent else "mlp")
    cspec = NetSpec(Ds, 64, 1, 1)
    return (GRUPPOLearner if recurrent else PPOLearner)("mappo", aspec, cspec, A, HParams(), dev)


@pytest.mark.parametrize("recurrent", [False, True])
def test_pinned_host_rollout_equals_host_collated_path(recurrent):
    """env workers' shared blocks page-locked, steps copied straight into the device rollout buffer and sampled
    there in place (cleanmarl_amd/host_rollout.py) == the host-collated shared-memory path (driver.host_rollout_shm), bit for bit:
    same envs, same actor, same Philox keys (all envs alive => same row indices)."""
    import torch
    from cleanmarl_amd.driver import HostActor, host_rollout_shm
    from cleanmarl_amd.env.shm_vector import ShmVectorEnv
    from cleanmarl_amd.host_rollout import PinnedHostRollout
    dev = torch.device("cuda:0")
    E, A, T = 12, 3, 9
    fac = dict(env_type="synthetic_cpu", env_name="x", env_family="mpe", agent_ids=True, kwargs={}, seed=5, synthetic=dict(agents=A, steps=T))
    L = _host_learner(A, 7 * A, 6 * A * A, 5, recurrent, dev)
    v1 = ShmVectorEnv(E, fac, n_workers=3)
    b1, s1 = host_rollout_shm(v1, HostActor(L, A, recurrent, dev, row_offset=7), E, A, 11, recurrent, dev)
    v1.close()
    v2 = ShmVectorEnv(E, fac, n_workers=3)
    pr = PinnedHostRollout(v2, L, recurrent, dev, row_offset=7, t_cap=4)  # t_cap 4 < T: the buffer grows twice
    b2, s2 = pr.collect(11)
    b3, s3 = pr.collect(12)  # a second episode through the same pinned blocks (other seed => other actions)
    pr.close(); v2.close()
    torch.cuda.synchronize()
    for k in ("obs", "state", "avail", "action", "logp", "reward", "ep_len"):
        assert torch.equal(getattr(b1, k), getattr(b2, k)), k
    assert s1["ep_len"] == s2["ep_len"] == [T] * E and s1["ep_reward"] == pytest.approx(s2["ep_reward"], abs=1e-5)
    assert b3.obs.shape == b2.obs.shape and b3.ep_len.tolist() == [T] * E and not torch.equal(b3.action, b2.action)


def test_pinned_host_rollout_ragged_episodes_replay():
    """Episodes of different lengths: the device-built batch is zero beyond each episode's end and, replayed on the CPU twin of the
    env with the recorded actions, reproduces every observation, state and reward."""
    import numpy as np
    import torch
    from cleanmarl_amd.env.shm_vector import ShmVectorEnv
    from cleanmarl_amd.env.synthetic import SyntheticSpreadEnv
    from cleanmarl_amd.host_rollout import PinnedHostRollout
    dev = torch.device("cuda:0")
    E, A, T = 10, 2, 11
    fac = dict(env_type="synthetic_cpu", env_name="x", env_family="mpe", agent_ids=True, kwargs={}, seed=3,
               synthetic=dict(agents=A, steps=T, ragged=True))
    L = _host_learner(A, 7 * A, 6 * A * A, 5, False, dev)
    v = ShmVectorEnv(E, fac, n_workers=4, index_offset=20)
    pr = PinnedHostRollout(v, L, False, dev, row_offset=20 * A)
    b, st = pr.collect(1)
    pr.close(); v.close()
    torch.cuda.synchronize()
    assert b.T == T and st["ep_len"] == [max(1, T - (20 + e) % 4) for e in range(E)] and b.ep_len.tolist() == st["ep_len"]
    for e in range(E):
        n = st["ep_len"][e]
        env = SyntheticSpreadEnv(A, True, max_cycles=n, seed=3, env_index=20 + e)
        o, _ = env.reset()
        for t in range(n):
            assert np.allclose(b.obs[e, :, t].cpu().numpy(), o) and np.allclose(b.state[e, t].cpu().numpy(), env.get_state())
            assert b.avail[e, :, t].all()
            o, r, d, tr, _ = env.step(b.action[e, :, t].cpu().numpy())
            assert abs(b.reward[e, t].item() - r) < 1e-5
        assert d or tr
        for k in ("obs", "avail", "action", "logp"):
            assert not getattr(b, k)[e, :, n:].any(), k
        assert not b.state[e, n:].any() and not b.reward[e, n:].any()
