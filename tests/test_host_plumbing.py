"""CPU tests of the host-side mirror of the reference interface: CLI defaults, pipe-protocol vector env,
rollout collation into the device layout, the C-ABI library's exported symbols (no compute calls)."""
import os
import re

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_cli_defaults_match_reference_scripts():
    from cleanmarl_amd.args import parse_args
    a = parse_args("mappo_multienvs", [])
    # cleanmarl/mappo_multienvs.py:18-79 (device differs on purpose: this build is GPU-only)
    exp = dict(env_type="smaclite", env_name="3m", env_family="mpe", agent_ids=True, batch_size=3, actor_hidden_dim=32,
               actor_num_layers=1, critic_hidden_dim=64, critic_num_layers=1, optimizer="Adam", learning_rate_actor=0.0008,
               learning_rate_critic=0.0008, total_timesteps=1000000, gamma=0.99, td_lambda=0.95, normalize_reward=False,
               normalize_advantage=False, normalize_return=False, epochs=3, ppo_clip=0.2, entropy_coef=0.001, clip_gradients=-1,
               log_every=10, eval_steps=50, num_eval_ep=10, use_wnb=False, wnb_project="", wnb_entity="", seed=1)
    for k, v in exp.items():
        assert getattr(a, k) == v, k
    assert parse_args("ippo_multienvs", []).critic_hidden_dim == 32            # cleanmarl/ippo_multienvs.py:34
    m = parse_args("mappo_lstm_multienvs", [])
    assert (m.tbptt, m.num_eval_ep, m.optimizer) == (10, 5, "Adam")            # cleanmarl/mappo_lstm_multienvs.py:64,70
    i = parse_args("ippo_lstm_multienvs", [])
    assert (i.tbptt, i.num_eval_ep, i.optimizer, i.critic_hidden_dim) == (5, 10, "AdamW", 32)  # ippo_lstm :38,66
    # single-environment front-ends: cleanmarl/mappo.py:65, ippo.py:33,65, mappo_lstm.py:61,69, ippo_lstm.py:33,37,65
    assert parse_args("mappo", []).eval_steps == 10 and parse_args("mappo", []).critic_hidden_dim == 64
    assert (parse_args("ippo", []).eval_steps, parse_args("ippo", []).critic_hidden_dim) == (50, 32)
    ml = parse_args("mappo_lstm", [])
    assert (ml.tbptt, ml.num_eval_ep, ml.eval_steps, ml.optimizer) == (10, 5, 50, "Adam")
    il = parse_args("ippo_lstm", [])
    assert (il.tbptt, il.num_eval_ep, il.optimizer, il.critic_hidden_dim) == (5, 10, "Adam", 32)
    # COMA: cleanmarl/coma_multienvs.py:19-89, coma.py:76-79
    c = parse_args("coma_multienvs", [])
    assert (c.critic_hidden_dim, c.learning_rate_actor, c.td_lambda, c.normalize_advantage, c.polyak, c.use_tdlamda, c.nsteps,
            c.start_e, c.end_e, c.exploration_fraction, c.target_network_update_freq, c.eval_steps, c.num_eval_ep) == \
        (128, 0.0005, 0.8, True, 0.005, True, 1, 0.5, 0.002, 750, 1, 10, 10)
    assert (parse_args("coma", []).eval_steps, parse_args("coma", []).num_eval_ep) == (50, 5)
    b = parse_args("mappo_multienvs", ["--env_type=pz", "--env-name", "simple_spread_v3", "--batch_size", "4",
                                       "--normalize_reward", "--no-agent_ids", "--clip_gradients=0.5", "--use_wnb=False"])
    assert (b.env_type, b.env_name, b.batch_size, b.normalize_reward, b.agent_ids, b.clip_gradients) == \
        ("pz", "simple_spread_v3", 4, True, False, 0.5)


def test_pipe_vector_env_and_collation():
    from cleanmarl_amd.driver import host_rollout
    from cleanmarl_amd.env.synthetic import SyntheticSpreadEnv
    from cleanmarl_amd.env.vector import PipeVectorEnv
    E, A, T = 3, 2, 5
    fac = dict(env_type="synthetic_cpu", env_name="x", env_family="mpe", agent_ids=True, kwargs={}, seed=3,
               synthetic=dict(agents=A, steps=T))
    venv = PipeVectorEnv(E, fac)
    assert venv.info() == {"obs_size": 7 * A, "action_size": 5, "n_agents": A, "state_size": 6 * A * A}
    rng = np.random.default_rng(0)
    script = rng.integers(0, 5, size=(T, E * A))

    class Stub:  # stands in for the on-device actor: deterministic scripted actions
        t = 0

        def act(self, obs, avail, h=None, seed=0, eps=0.0):
            a = script[self.t][:obs.shape[0] * A]
            self.t += 1
            return a.astype(np.int32), np.full(a.shape, -1.5, np.float32), None

    b, stats = host_rollout(venv, Stub(), E, A, 0, False, torch.device("cpu"))
    venv.close()
    assert b.obs.shape == (E, A, T, 7 * A) and b.ep_len.tolist() == [T] * E and stats["ep_len"] == [T] * E
    for e in range(E):
        env = SyntheticSpreadEnv(A, True, max_cycles=T, seed=3, env_index=e)
        o, _ = env.reset()
        tot = 0.0
        for t in range(T):
            assert np.allclose(b.obs[e, :, t].numpy(), o) and np.allclose(b.state[e, t].numpy(), env.get_state())
            acts = script[t][e * A:(e + 1) * A]
            assert b.action[e, :, t].tolist() == list(acts)
            o, r, d, tr, _ = env.step(acts)
            assert abs(b.reward[e, t].item() - r) < 1e-6
            tot += r
        assert abs(stats["ep_reward"][e] - tot) < 1e-5
    assert b.avail.all() and (b.logp == -1.5).all()


def test_shm_vector_env_matches_pipe_protocol_and_is_faster():
    """the shared-memory batched-step vector env collects the same batch as the reference-style
    pipe-per-env protocol (ragged episode lengths included) with fewer host round trips."""
    import time
    from cleanmarl_amd.driver import host_rollout, host_rollout_shm
    from cleanmarl_amd.env.shm_vector import ShmVectorEnv
    from cleanmarl_amd.env.vector import PipeVectorEnv
    E, A, T = 24, 3, 12
    fac = dict(env_type="synthetic_cpu", env_name="x", env_family="mpe", agent_ids=True, kwargs={}, seed=5,
               synthetic=dict(agents=A, steps=T, ragged=True))

    class Stub:  # deterministic actor: the action is a function of the observation only
        def act(self, obs, avail, h=None, seed=0, eps=0.0):
            a = (np.abs(obs[..., :4]).sum(-1) * 1000).astype(np.int64) % 5
            return a.reshape(-1).astype(np.int32), np.full(a.size, -0.7, np.float32), None

    pv = PipeVectorEnv(E, fac)
    t0 = time.perf_counter(); b1, s1 = host_rollout(pv, Stub(), E, A, 0, False, torch.device("cpu")); t_pipe = time.perf_counter() - t0
    pv.close()
    sv = ShmVectorEnv(E, fac, n_workers=4)
    t0 = time.perf_counter(); b2, s2 = host_rollout_shm(sv, Stub(), E, A, 0, False, torch.device("cpu")); t_shm = time.perf_counter() - t0
    sv.close()
    for k in ("obs", "state", "avail", "action", "logp", "reward", "ep_len"):
        assert torch.equal(getattr(b1, k), getattr(b2, k)), k
    assert s1["ep_len"] == s2["ep_len"] and len(set(s1["ep_len"])) > 1 and np.allclose(s1["ep_reward"], s2["ep_reward"], atol=1e-5)
    assert t_shm < t_pipe * 1.5  # not a benchmark (24 tiny envs); tools/bench_host_env.py measures the real gap


def test_library_exports_every_declared_symbol():
    """include/cleanmarl_hip.h <-> libcleanmarl_hip.so <-> ctypes table stay in sync (no GPU needed)."""
    from cleanmarl_amd import _native
    from cleanmarl_amd.build import build_native
    build_native()  # no-op when up to date; hipcc cross-compiles for gfx950 without a GPU
    hdr = open(os.path.join(ROOT, "include", "cleanmarl_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(cm_[a-z0-9_]+)\s*\(", hdr))
    assert declared, "no declarations parsed"
    lib = _native.load()
    for name in sorted(declared):
        assert hasattr(lib, name), f"{name} declared in the header but not exported"
        assert name in _native.SIGNATURES, f"{name} has no ctypes signature"
    assert set(_native.SIGNATURES) == declared
    assert lib.cm_version() >= 100
    assert lib.cm_mlp_param_count(56, 64, 1, 5) == 56 * 64 + 64 + 64 * 64 + 64 + 5 * 64 + 5


def test_rank_shards_of_host_vector_envs_are_distinct_and_tile_the_global_batch():
    """with the batch env-sharded over ranks, rank r's vector env must own the GLOBAL env indices
    [env_offset, env_offset + E_local) -- not 0..E_local-1 again -- for both the pipe and the shared-memory flavour."""
    from cleanmarl_amd.env.shm_vector import ShmVectorEnv
    from cleanmarl_amd.env.vector import PipeVectorEnv
    A, T = 2, 4
    fac = dict(env_type="synthetic_cpu", env_name="x", env_family="mpe", agent_ids=True, kwargs={}, seed=5,
               synthetic=dict(agents=A, steps=T))
    full = PipeVectorEnv(4, fac)
    want = np.stack([c["obs"] for c in full.reset_all()])
    full.close()
    assert not np.allclose(want[:2], want[2:])  # index-keyed envs: different indices give different episodes
    for off in (0, 2):
        p = PipeVectorEnv(2, fac, index_offset=off)
        got = np.stack([c["obs"] for c in p.reset_all()])
        p.close()
        assert np.allclose(got, want[off:off + 2])
        s = ShmVectorEnv(2, fac, n_workers=2, index_offset=off)
        s._all("reset")
        got = s.arr["obs"].copy()
        s.close()
        assert np.allclose(got, want[off:off + 2])


def test_hand_issued_lds_reads_are_never_touched_in_flight():
    """csrc/cm_critic_fused.h and csrc/cm_gru_step2.h issue LDS reads through inline asm and wait for them by hand; the compiler
    does not know they are asynchronous.  The generated assembly must not read or overwrite their destination registers, branch or
    reach a label before the covering s_waitcnt (tools/lint_lds_hazards.py; a phi copy at a branch once did)."""
    from cleanmarl_amd.build import lint_hand_pipelines
    reads, kernels = lint_hand_pipelines()
    assert reads >= 1000 and kernels >= 50


def test_lds_hazard_linter_flags_a_copy_before_the_wait(tmp_path):
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location("lint", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "lint_lds_hazards.py"))
    lint = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(lint)
    good = "k:\n\t;;#ASMSTART\n\tds_read_b128 v[4:7], v1 offset:0\n\t;;#ASMEND\n\tv_add_f32_e32 v9, v8, v8\n\t;;#ASMSTART\n\ts_waitcnt lgkmcnt(0)\n\t;;#ASMEND\n\tv_mov_b32_e32 v2, v4\n\ts_endpgm\n"
    bad = good.replace("v_add_f32_e32 v9, v8, v8", "v_mov_b64_e32 v[10:11], v[4:5]")
    branch = good.replace("v_add_f32_e32 v9, v8, v8", "s_cbranch_vccz .LBB0_1")
    for text, want in ((good, 0), (bad, 1), (branch, 1)):
        f = tmp_path / "k.s"
        f.write_text(text)
        probs, nk, nhand = lint.lint_file(str(f))
        assert nk == 1 and nhand == 1 and len(probs) == want, (text, probs)


@pytest.mark.parametrize("kind", ["pipe", "shm"])
def test_a_dead_env_worker_is_an_error_not_a_hang(kind):
    """The reference waits in a bare recv() (cleanmarl/mappo_multienvs.py:318, 396) and keeps the child's pipe end open in the parent: an
    env process that dies leaves the run hanging for ever.  Both vector envs of the build notice it (the parent closes the child's end
    after the fork and polls with a liveness check) and raise EnvWorkerDied."""
    import os
    import signal
    from cleanmarl_amd.env.shm_vector import ShmVectorEnv
    from cleanmarl_amd.env.vector import EnvWorkerDied, PipeVectorEnv
    fac = dict(env_type="synthetic_cpu", env_name="x", env_family="mpe", agent_ids=True, kwargs={}, seed=3, synthetic=dict(agents=2, steps=5, ragged=False))
    if kind == "pipe":
        v = PipeVectorEnv(3, fac)
        v.reset_all()
        os.kill(v.procs[1].pid, signal.SIGKILL)
        v.procs[1].join(timeout=5)
        with pytest.raises(EnvWorkerDied, match="env worker 1"):
            v.reset_all()
    else:
        v = ShmVectorEnv(4, fac, n_workers=2)
        v._all("reset")
        os.kill(v.procs[0].pid, signal.SIGKILL)
        v.procs[0].join(timeout=5)
        with pytest.raises(EnvWorkerDied, match="worker 0"):
            v._all("reset")
    try:
        v.close()
    except Exception:  # noqa: BLE001 -- closing a half-dead pool must not raise anything new; a worker that is gone is gone
        pass
