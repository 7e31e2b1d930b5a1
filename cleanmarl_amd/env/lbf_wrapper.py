"""Level-Based Foraging adapter (optional dependency, imported lazily) -- the third env type of the COMA scripts.

Contract taken from the reference's adapter (cleanmarl/env/lbf.py:9-76): a gymnasium ``TimeLimit`` of 150 steps around
``gym.make(map_name)``, the per-agent rewards aggregated to ONE team reward (sum by default, or mean), the global state =
the concatenation of all agents' observations, observations optionally extended with a one-hot agent id, availability masks
= ones for an agent's own actions and zeros up to the widest action space, and ``truncated`` forced when the env reports
termination exactly at the step limit.
"""
import numpy as np

from .common_interface import CommonInterface


def _require_lbf():
    try:
        import gymnasium
        from gymnasium.spaces import flatdim
        from gymnasium.wrappers import TimeLimit
        import lbforaging  # noqa: F401  -- importing registers the Foraging-* ids
    except ImportError as exc:  # pragma: no cover - optional dependency
        raise ImportError(f"env_type=lbf needs the 'lbforaging' and 'gymnasium' packages ({exc})") from exc
    return gymnasium, TimeLimit, flatdim


class LBFWrapper(CommonInterface):
    def __init__(self, map_name, reward_aggr="sum", seed=0, time_limit=150, agent_ids=False, **make_kwargs):
        gymnasium, TimeLimit, flatdim = _require_lbf()
        if reward_aggr not in ("sum", "mean"):
            raise ValueError(f"reward_aggr={reward_aggr!r}: expected 'sum' or 'mean'")
        self.env = TimeLimit(gymnasium.make(map_name, max_episode_steps=time_limit, **make_kwargs), max_episode_steps=time_limit)
        self.core = self.env.unwrapped
        self.episode_limit = int(time_limit)
        self.agent_ids, self.reward_aggr = bool(agent_ids), reward_aggr
        self.n_agents = int(self.core.n_agents)
        self._id_block = np.eye(self.n_agents)
        self._act_dims = [int(flatdim(sp)) for sp in self.env.action_space]
        self._n_actions = max(self._act_dims)
        self._obs_dim = max(int(flatdim(sp)) for sp in self.env.observation_space)
        self._t = 0
        self._state = np.zeros(self._obs_dim * self.n_agents)

    def _observe(self, raw):
        raw = np.asarray(raw)
        self._state = raw.reshape(-1)
        return np.concatenate((raw, self._id_block), axis=1) if self.agent_ids else raw

    def reset(self, seed=None):
        self._t = 0
        raw, _info = self.env.reset(seed=seed)
        return self._observe(raw), {}

    def step(self, actions):
        raw, rewards, terminated, truncated, info = self.env.step([int(a) for a in actions])
        self._t += 1
        team = np.sum(rewards) if self.reward_aggr == "sum" else np.mean(rewards)
        if terminated and self._t == getattr(self.core, "_max_episode_steps", self.episode_limit):
            truncated = True
        return self._observe(raw), np.array(team), terminated, truncated, info

    def get_avail_actions(self):
        mask = np.zeros((self.n_agents, self._n_actions), dtype=np.int64)
        for a, n in enumerate(self._act_dims):
            mask[a, :n] = 1
        return mask

    def get_state(self):
        return self._state

    def get_obs_size(self):
        return self._obs_dim + (self.n_agents if self.agent_ids else 0)

    def get_state_size(self):
        return self._obs_dim * self.n_agents

    def get_action_size(self):
        return self._n_actions

    def sample(self):
        return list(self.env.action_space.sample())

    def close(self):
        return self.env.close()
