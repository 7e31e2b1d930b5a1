"""Batched evaluation (cleanmarl_amd/evaluate.py <- cleanmarl/mappo_multienvs.py:614-650, mappo_lstm_multienvs.py:675-714).

GPU: the device evaluator (num_eval_ep episodes as ONE rollout on the evaluation stream) and the host evaluator (the CPU twins of the
envs stepped side by side, one act call per time step) draw from the same Philox keys, so they must play the SAME episodes -- sampled
and greedy, MLP and GRU actors, spread and shape envs.  CPU: the host evaluator's episode dealing over two gloo ranks reproduces the
one-process result in slot order.
"""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from cleanmarl_amd.args import parse_args


def _setup(script, env_type, extra=()):
    from cleanmarl_amd.env.vector import environment
    from cleanmarl_amd.evaluate import eval_base
    from cleanmarl_amd.learner import HParams, NetSpec, PPOLearner, init_params_like_torch
    argv = [f"--env_type={env_type}", "--batch_size=8", "--synthetic_agents=3", "--synthetic_steps=21", "--num_eval_ep=5",
            "--seed=3"] + list(extra)
    args = parse_args(script, argv)
    recurrent = "lstm" in script
    synth = dict(agents=args.synthetic_agents, steps=args.synthetic_steps, obs=args.synthetic_obs, state=args.synthetic_state,
                 actions=args.synthetic_actions, avail_p=args.synthetic_avail_p)
    fac = dict(env_type=env_type + "_cpu", env_name=args.env_name, env_family=args.env_family, agent_ids=args.agent_ids, kwargs={},
               seed=args.seed, synthetic=synth)
    first = environment(**dict(fac, index=eval_base(args.batch_size)))
    A, Do, Ds, K = first.n_agents, first.get_obs_size(), first.get_state_size(), first.get_action_size()
    dev = torch.device("cuda:0")
    aspec = NetSpec(Do, 64, 0 if recurrent else 1, K, "gru" if recurrent else "mlp")
    cspec = NetSpec(Ds, 64, 1, 1)
    torch.manual_seed(11)
    ap, cp = init_params_like_torch(aspec), init_params_like_torch(cspec)
    ap = [p * 3.0 for p in ap]  # peaked policies: greedy and sampled episodes differ visibly
    if recurrent:
        from cleanmarl_amd.gru import GRUPPOLearner
        L = GRUPPOLearner("mappo", aspec, cspec, A, HParams(), dev, ap, cp)
    else:
        L = PPOLearner("mappo", aspec, cspec, A, HParams(), dev, ap, cp)
    return args, fac, first, L, aspec, A, dev, recurrent


@pytest.mark.gpu
@pytest.mark.parametrize("greedy", [False, True])
@pytest.mark.parametrize("script,env_type,extra", [
    ("mappo_multienvs", "synthetic", ()),
    ("mappo_lstm_multienvs", "synthetic", ()),
    ("ippo_multienvs", "synthetic_shape", ("--synthetic_obs=33", "--synthetic_state=50", "--synthetic_actions=17")),
    ("ippo_lstm_multienvs", "synthetic_shape", ("--synthetic_obs=33", "--synthetic_state=50", "--synthetic_actions=17")),
])
def test_device_eval_plays_the_host_evaluators_episodes(script, env_type, extra, greedy):
    from cleanmarl_amd.driver import HostActor
    from cleanmarl_amd.env.vector import environment
    from cleanmarl_amd.evaluate import DeviceEvaluator, HostEvaluator
    args, fac, first, L, aspec, A, dev, recurrent = _setup(script, env_type, extra)
    de = DeviceEvaluator(args, aspec, A, dev, args.batch_size)
    he = HostEvaluator(lambda index: environment(**dict(fac, index=index)), first, HostActor(L, A, recurrent, dev), args, A, recurrent,
                       dev, args.batch_size)
    he.record = True
    seen = []
    for rnd in (1, 2, 7):  # evaluation round n = episode n of the evaluation envs
        d = de.launch(L.actor, rnd, greedy=greedy)
        h = he.run(rnd, greedy=greedy)
        b = de.roll.batch
        torch.cuda.synchronize()
        dev_act = b.action.permute(2, 0, 1).cpu().numpy()  # [T, E, A]
        host_act = np.stack(he.actions)
        assert np.array_equal(dev_act, host_act), (script, rnd, greedy)
        assert d.ep_lengths == h.ep_lengths == [float(args.synthetic_steps)] * 5 or d.ep_lengths == [args.synthetic_steps] * 5
        np.testing.assert_allclose(d.ep_rewards, h.ep_rewards, rtol=2e-6, atol=2e-5)
        seen.append(tuple(np.round(d.ep_rewards, 4)))
    assert len(set(seen)) == 3  # different rounds are different episodes
    he.close()


@pytest.mark.gpu
def test_greedy_and_sampled_eval_differ_and_training_sampler_is_untouched():
    from cleanmarl_amd.driver import HostActor
    from cleanmarl_amd.evaluate import DeviceEvaluator
    args, fac, first, L, aspec, A, dev, recurrent = _setup("mappo_multienvs", "synthetic")
    de = DeviceEvaluator(args, aspec, A, dev, args.batch_size)
    s = de.launch(L.actor, 2, greedy=False).ep_rewards
    g = de.launch(L.actor, 2, greedy=True).ep_rewards
    s2 = de.launch(L.actor, 2, greedy=False).ep_rewards
    assert s == s2 and s != g
    ha = HostActor(L, A, recurrent, dev)
    x = np.zeros((2, A, aspec.din), np.float32); av = np.ones((2, A, aspec.dout), np.int64)
    ha.act(x, av, seed=1, t=5, row_offset=77)   # an evaluator's call: explicit keys
    assert ha.calls == 0                        # ... leaves the training rollouts' call counter alone
    ha.act(x, av, seed=1)
    assert ha.calls == 1


@pytest.mark.gpu
def test_cli_logs_deferred_device_eval_at_the_step_it_was_launched(tmp_path, monkeypatch):
    """eval/* of iteration i are read from the pinned slot one iteration later but carry iteration i's env-step x value, once per
    evaluation round, in the reference's tag set; --greedy_eval and the default cadence arithmetic ((training_step / epochs) % eval_steps)."""
    from cleanmarl_amd.driver import run
    monkeypatch.chdir(tmp_path)
    for extra in ([], ["--greedy_eval"]):
        out = run("mappo_multienvs", ["--env_type=synthetic", "--batch_size=4", "--synthetic_agents=3", "--synthetic_steps=25",
                                      "--total_timesteps=600", "--eval_steps=2", "--num_eval_ep=3", "--log_every=1"] + extra)
        ev = [(t, s) for t, _, s in out["history"] if t.startswith("eval/")]
        assert [s for t, s in ev if t == "eval/ep_reward"] == [200, 400, 600]
        assert {t for t, _ in ev} == {"eval/ep_reward", "eval/std_ep_reward", "eval/ep_length"}
        assert [v for t, v, _ in out["history"] if t == "eval/ep_length"] == [25.0] * 3


# ---------------------------------------------------------------------------------------------------------------- CPU (gloo, world 2)
class _FakeActor:
    """Deterministic stand-in for driver.HostActor: the action of a row depends on its Philox-style key (seed, global row, t) only."""

    def act(self, obs, avail, h=None, seed=0, greedy=False, eps=0.0, t=None, row_offset=None):
        rows = obs.shape[0] * obs.shape[1]
        r = np.arange(rows, dtype=np.int64) + int(row_offset)
        a = (r * 7 + int(t) * 3 + (int(seed) % 1000)) % 5
        return a.astype(np.int32), np.zeros(rows, np.float32), None


def _host_eval(rank, world, n_ep=5, live=0):
    from cleanmarl_amd.env.vector import environment
    from cleanmarl_amd.evaluate import HostEvaluator, eval_base
    args = parse_args("mappo_multienvs", ["--env_type=synthetic_cpu", "--batch_size=8", "--synthetic_agents=3", "--synthetic_steps=9",
                                          f"--num_eval_ep={n_ep}", "--seed=5", f"--eval_live_envs={live}"])
    synth = dict(agents=3, steps=9, ragged=True)
    fac = dict(env_type="synthetic_cpu", env_name="x", env_family="mpe", agent_ids=True, kwargs={}, seed=5, synthetic=synth)
    first = environment(**dict(fac, index=eval_base(8)))
    he = HostEvaluator(lambda index: environment(**dict(fac, index=index)), first, _FakeActor(), args, 3, False, torch.device("cpu"), 8,
                       rank, world, None)
    out = []
    for rnd in (0, 1):
        r = he.run(rnd)
        out.append((r.ep_rewards, r.ep_lengths))
    he.close()
    return out


def _eval_worker(rank, world, port, path, n_ep=5):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    torch.distributed.init_process_group("gloo", rank=rank, world_size=world)
    out = _host_eval(rank, world, n_ep)
    torch.save(out, f"{path}.{rank}")
    torch.distributed.destroy_process_group()


def test_host_evaluator_deals_episodes_over_ranks_and_gathers_in_slot_order(tmp_path):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    path = str(tmp_path / "ev")
    mp.spawn(_eval_worker, args=(2, port, path), nprocs=2, join=True)
    one = _host_eval(0, 1)
    r0, r1 = torch.load(f"{path}.0"), torch.load(f"{path}.1")
    assert r0 == r1  # every rank holds the complete result
    for (ra, la), (rb, lb) in zip(one, r0):
        assert la == lb and len(la) == 5 and len(set(la)) > 1  # ragged horizons (index % 4), slot order kept
        np.testing.assert_allclose(ra, rb, rtol=0, atol=0)


def test_host_evaluator_deals_ten_episodes_over_eight_ranks(tmp_path):
    """The reference's default num_eval_ep = 10 on the eight ranks of a node: blocks of ceil(10 / 8) = 2 slots, so ranks 5 - 7 play NO
    episode -- they still enter the gather, and every rank ends up with the ten results of the one-process evaluation, in slot order."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    path = str(tmp_path / "ev8")
    mp.spawn(_eval_worker, args=(8, port, path, 10), nprocs=8, join=True)
    one = _host_eval(0, 1, 10)
    got = [torch.load(f"{path}.{r}") for r in range(8)]
    assert all(g == got[0] for g in got)
    for (ra, la), (rb, lb) in zip(one, got[0]):
        assert la == lb and len(la) == 10
        np.testing.assert_allclose(ra, rb, rtol=0, atol=0)


def test_host_evaluator_waves_bound_the_live_envs_and_play_the_same_episodes():
    """--eval_live_envs: envs are created on first use, at most N alive at once (waves), N = 1 = the reference's sequential evaluation;
    the action keys do not depend on the wave size, so every setting returns the episodes of the all-at-once evaluation."""
    from cleanmarl_amd.env.vector import environment
    from cleanmarl_amd.evaluate import HostEvaluator, eval_base
    full = _host_eval(0, 1, 7, live=0)
    for live in (1, 3, 7, 9):
        assert _host_eval(0, 1, 7, live=live) == full, live
    args = parse_args("mappo_multienvs", ["--env_type=synthetic_cpu", "--batch_size=8", "--synthetic_agents=3", "--synthetic_steps=9",
                                          "--num_eval_ep=7", "--seed=5", "--eval_live_envs=2"])
    fac = dict(env_type="synthetic_cpu", env_name="x", env_family="mpe", agent_ids=True, kwargs={}, seed=5, synthetic=dict(agents=3, steps=9, ragged=True))
    made, closed = [], []

    def make(index):
        e = environment(**dict(fac, index=index))
        made.append(index)
        orig = e.close
        e.close = lambda: (closed.append(index), orig())[1]
        return e
    he = HostEvaluator(make, environment(**dict(fac, index=eval_base(8))), _FakeActor(), args, 3, False, torch.device("cpu"), 8)
    assert he.envs == {} and made == []  # nothing is built before the first round
    live_max = [0]
    play = he._play
    he._play = lambda slots, *a: (live_max.__setitem__(0, max(live_max[0], len(he.envs) + sum(1 for j in slots if j not in he.envs))), play(slots, *a))[1]
    he.run(0)
    assert live_max[0] <= 2 and len(made) == 6 and len(closed) == 6  # slots 1..6 built and closed again; slot 0 is the caller's env
    he.close()


class _StatefulEnv:
    """A heavy-env stand-in WITHOUT counter-keyed episodes (no `episode` attribute): its RNG advances from episode to episode, like the
    reference's single eval env (cleanmarl/mappo_multienvs.py:614-650)."""

    def __init__(self, index, log):
        self.rng, self.t, self.log, self.index = np.random.default_rng(index), 0, log, index
        log.append(("make", index))

    def reset(self, seed=None):
        self.t, self.h = 0, int(self.rng.integers(3, 7))
        self.log.append(("reset", self.index))
        return np.zeros((3, 4), np.float32), {}

    def get_avail_actions(self):
        return np.ones((3, 5), np.int64)

    def step(self, actions):
        self.t += 1
        return np.zeros((3, 4), np.float32), float(self.rng.random()), False, self.t >= self.h, {}

    def close(self):
        self.log.append(("close", self.index))


def test_host_evaluator_pools_heavy_envs_across_waves_and_rounds():
    """--eval_live_envs=N with envs that are not counter-keyed: at most N envs are EVER built (wave position i reuses pool env i, reset per
    episode), nothing is rebuilt between waves or rounds, and the envs' RNG streams advance across rounds -- with N = 1 that is the
    reference's sequential evaluation on one env (the per-wave close / rebuild paid env construction per episode per round and
    restarted every slot from the same freshly seeded state)."""
    from cleanmarl_amd.evaluate import HostEvaluator, eval_base
    for live, n_ep in ((1, 4), (2, 5)):
        args = parse_args("mappo_multienvs", ["--env_type=synthetic_cpu", "--batch_size=8", f"--num_eval_ep={n_ep}", "--seed=5", f"--eval_live_envs={live}"])
        log = []
        first = _StatefulEnv(eval_base(8), log)
        he = HostEvaluator(lambda index: _StatefulEnv(index, log), first, _FakeActor(), args, 3, False, torch.device("cpu"), 8)
        r0 = he.run(0)
        r1 = he.run(1)
        made = [i for k, i in log if k == "make"]
        assert len(made) == live and not [1 for k, _ in log if k == "close"]  # the caller's env + (live - 1) pool envs, nothing closed or rebuilt
        assert sum(1 for k, _ in log if k == "reset") == 2 * n_ep  # one reset per episode
        assert r0.ep_rewards != r1.ep_rewards  # the RNG streams advanced: round 1 is not a replay of round 0
        he.close()
        assert sorted(i for k, i in log if k == "close") == sorted(made[1:])  # the pool is closed once, the caller's env is left alone
