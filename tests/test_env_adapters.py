"""The PettingZoo / SMAClite adapters executed against stand-in packages (tests/stub_envs.py): adapter contract of
cleanmarl/env/pettingzoo_wrapper.py:33-101 and cleanmarl/env/smaclite_wrapper.py:12-60, then BASELINE config 1's plumbing
(`--env_type=pz --env_family=mpe --env_name=simple_spread_v3`, 4 envs) through both vector envs and the collation."""
import numpy as np
import torch

from stub_envs import install


def test_pettingzoo_adapter_contract(monkeypatch):
    install(monkeypatch)
    from cleanmarl_amd.env.vector import environment
    env = environment("pz", "simple_spread_v3", "mpe", True, kwargs=dict(N=3, max_cycles=4))
    assert (env.n_agents, env.get_obs_size(), env.get_state_size(), env.get_action_size()) == (3, 18 + 3, 18 * 3, 5)
    obs, info = env.reset(seed=2)
    assert obs.shape == (3, 21) and info == {}
    assert np.array_equal(obs[:, 18:], np.eye(3))                          # one-hot agent ids appended (:96-97)
    assert np.array_equal(env.get_state(), obs[:, :18].reshape(-1))        # state = concat of the RAW observations (:94-95)
    assert np.array_equal(env.get_avail_actions(), np.ones((3, 5), int))   # every action legal (:79-90)
    for t in range(1, 5):
        obs, r, done, trunc, info = env.step(np.array([0, 1, 2]))
        assert r == -1.0 * t                                               # the FIRST agent's reward is the team reward (:66)
        assert done is False and trunc == (t == 4)                         # truncation at max_cycles, observations still real
        assert info["agent_1_t"] == t and np.array_equal(env.get_state(), obs[:, :18].reshape(-1))
    assert len(env.sample()) == 3
    env.close()
    # no ids: obs == raw rows
    raw = environment("pz", "simple_spread_v3", "mpe", False, kwargs=dict(N=2))
    assert raw.get_obs_size() == 18 and raw.reset()[0].shape == (2, 18)


def test_pettingzoo_adapter_reuses_the_last_observation_on_termination(monkeypatch):
    """A terminated PettingZoo episode returns EMPTY dicts; the adapter hands back the last real observation and zero reward
    (cleanmarl/env/pettingzoo_wrapper.py:58-64)."""
    install(monkeypatch)
    from cleanmarl_amd.env.vector import environment
    env = environment("pz", "simple_spread_v3", "mpe", True, kwargs=dict(N=3, terminate_at=3))
    env.reset()
    env.step([1, 1, 1])
    last, _, done, _, _ = env.step([2, 0, 4])
    assert not done
    obs, r, done, trunc, info = env.step([0, 0, 0])
    assert done is True and trunc is False and r == 0 and info == {}
    assert np.array_equal(obs, last) and np.array_equal(env.get_state(), last[:, :18].reshape(-1))


def test_pettingzoo_adapter_pads_availability_to_the_longest_action_space(monkeypatch):
    install(monkeypatch)
    from cleanmarl_amd.env.vector import environment
    env = environment("pz", "simple_spread_v3", "mpe", False, kwargs=dict(N=3, n_actions=[3, 5, 4]))
    assert env.get_avail_actions().tolist() == [[1, 1, 1, 0, 0], [1, 1, 1, 1, 1], [1, 1, 1, 1, 0]]
    assert env.get_action_size() == 3  # the reference reports agent 0's action count (:77-78)


def test_smaclite_adapter_contract(monkeypatch):
    install(monkeypatch)
    from cleanmarl_amd.env.vector import environment
    env = environment("smaclite", "3m", "mpe", True)
    assert (env.n_agents, env.get_obs_size(), env.get_state_size(), env.get_action_size(), env.episode_limit) == (3, 33, 48, 9, 150)
    obs, info = env.reset()
    assert obs.shape == (3, 33) and info == {} and np.array_equal(obs[:, 30:], np.eye(3))
    t = 0
    while True:
        av = env.get_avail_actions()
        assert av.shape == (3, 9) and av[:, 1 + t % 8].sum() == 0 and av.sum() == 24   # masks come from the env itself
        acts = env.sample()
        assert all(av[i, a] for i, a in enumerate(acts))                                # sample() only draws legal actions
        obs, r, term, trunc, info = env.step(np.asarray(acts))
        t += 1
        assert r == float(sum(acts)) and not term                                       # the env's TEAM reward, unchanged
        assert np.array_equal(env.get_state(), np.full(48, float(t), np.float32))
        if trunc:
            break
    assert t == 150                                                                     # gymnasium TimeLimit(150) (:13-14)
    short = environment("smaclite", "3m", "mpe", False, kwargs=dict(time_limit=7, win_at=5))
    o, _ = short.reset()
    assert o.shape == (3, 30)
    for k in range(5):
        o, r, term, trunc, info = short.step([0, 0, 0])
    assert term and not trunc and info["battle_won"] is True


def _scripted_actor(A, K):
    class Stub:  # stands in for the on-device actor: first legal action
        def act(self, obs, avail, h=None, seed=0, eps=0.0):
            a = np.argmax(np.asarray(avail).reshape(-1, K), axis=1).astype(np.int32)
            return a, np.full(a.shape, -1.0, np.float32), None
    return Stub()


def test_config1_plumbing_pz_simple_spread_four_envs(monkeypatch):
    """BASELINE.json configs[0]: `mappo_multienvs.py --env_type=pz --env_family=mpe --env_name=simple_spread_v3`, 4 envs, through the
    reference's pipe protocol AND the shared-memory vector env; both collate to the same device-layout batch."""
    install(monkeypatch)
    from cleanmarl_amd.driver import host_rollout, host_rollout_shm
    from cleanmarl_amd.env.shm_vector import ShmVectorEnv
    from cleanmarl_amd.env.vector import PipeVectorEnv
    E, A, T = 4, 3, 25
    fac = dict(env_type="pz", env_name="simple_spread_v3", env_family="mpe", agent_ids=True, kwargs={}, seed=1, synthetic=None)
    pv = PipeVectorEnv(E, fac)
    assert pv.info() == {"obs_size": 21, "action_size": 5, "n_agents": 3, "state_size": 54}
    b1, s1 = host_rollout(pv, _scripted_actor(A, 5), E, A, 0, False, torch.device("cpu"))
    pv.close()
    sv = ShmVectorEnv(E, fac, n_workers=2)
    b2, s2 = host_rollout_shm(sv, _scripted_actor(A, 5), E, A, 0, False, torch.device("cpu"))
    sv.close()
    assert b1.obs.shape == (E, A, T, 21) and b1.state.shape == (E, T, 54) and b1.ep_len.tolist() == [T] * E
    for k in ("obs", "state", "avail", "action", "logp", "reward", "ep_len"):
        assert torch.equal(getattr(b1, k), getattr(b2, k)), k
    assert s1["ep_len"] == s2["ep_len"] == [T] * E
    assert abs(s1["ep_reward"][0] + sum(range(1, T + 1))) < 1e-6  # sum of agent 0's rewards -t
    assert torch.equal(b1.obs[0, :, 0, 18:], torch.eye(3))


def test_smaclite_plumbing_ragged_battles(monkeypatch):
    install(monkeypatch)
    from cleanmarl_amd.driver import host_rollout_shm
    from cleanmarl_amd.env.shm_vector import ShmVectorEnv
    fac = dict(env_type="smaclite", env_name="3m", env_family="mpe", agent_ids=True, kwargs=dict(time_limit=12, win_at=9), seed=1,
               synthetic=None)
    sv = ShmVectorEnv(3, fac, n_workers=2)
    b, stats = host_rollout_shm(sv, _scripted_actor(3, 9), 3, 3, 0, False, torch.device("cpu"))
    sv.close()
    assert b.ep_len.tolist() == [9, 9, 9] and b.avail.shape == (3, 3, 9, 9) and b.obs.shape == (3, 3, 9, 33)
    assert all(i["battle_won"] for i in stats["infos"])  # what driver.py logs as rollout/battle_won
    assert int(b.avail[0, 0, 0].sum()) == 8 and int(b.avail[0, 0, 0, 1]) == 0
