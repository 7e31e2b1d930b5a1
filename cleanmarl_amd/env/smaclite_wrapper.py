"""SMAClite adapter (optional dependency, imported lazily).

Contract taken from the reference's adapter (cleanmarl/env/smaclite_wrapper.py:12-60): episodes are cut by a
gymnasium ``TimeLimit`` of 150 steps, per-agent observations optionally get a one-hot agent id appended, the global
state and the availability masks come from the underlying env, the reward is the env's team reward.
"""
import numpy as np

from .common_interface import CommonInterface


def _require_smaclite():
    try:
        import gymnasium
        from gymnasium.wrappers import TimeLimit
        import smaclite  # noqa: F401  -- importing registers "smaclite/<map>-v0"
    except ImportError as exc:  # pragma: no cover - optional dependency
        raise ImportError("env_type=smaclite needs the 'smaclite' and 'gymnasium' packages "
                          f"({exc}); the bundled --env_type=synthetic_shape has the same tensor shapes") from exc
    return gymnasium, TimeLimit


class SMACliteWrapper(CommonInterface):
    def __init__(self, map_name, seed=0, time_limit=150, agent_ids=False, **make_kwargs):
        gymnasium, TimeLimit = _require_smaclite()
        base = gymnasium.make(f"smaclite/{map_name}-v0", seed=seed, **make_kwargs)
        self.env = TimeLimit(base, max_episode_steps=time_limit)
        self.core = self.env.unwrapped
        self.episode_limit = int(time_limit)
        self.agent_ids = bool(agent_ids)
        self.n_agents = int(self.core.n_agents)
        self._id_block = np.eye(self.n_agents)
        self._n_actions = max(int(space.n) for space in self.env.action_space)

    # ---- stepping
    def _with_ids(self, per_agent_obs):
        per_agent_obs = np.asarray(per_agent_obs)
        if not self.agent_ids:
            return per_agent_obs
        return np.concatenate((per_agent_obs, self._id_block), axis=1)

    def reset(self, seed=None, options=None):
        first_obs, _info = self.env.reset(seed=seed, options=options)
        return self._with_ids(first_obs), {}

    def step(self, actions):
        next_obs, team_reward, terminated, truncated, info = self.env.step([int(a) for a in actions])
        return self._with_ids(next_obs), team_reward, terminated, truncated, info

    # ---- queries
    def get_avail_actions(self):
        return np.asarray(self.core.get_avail_actions())

    def get_state(self):
        return self.core.get_state()

    def get_obs_size(self):
        return int(self.core.obs_size) + (self.n_agents if self.agent_ids else 0)

    def get_state_size(self):
        return int(self.core.state_size)

    def get_action_size(self):
        return self._n_actions

    def sample(self):
        legal = np.asarray(self.get_avail_actions(), dtype=np.float64)
        return [int(np.random.choice(legal.shape[1], p=row / row.sum())) for row in legal]

    def close(self):
        self.env.close()
