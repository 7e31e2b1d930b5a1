"""SMAClite adapter implementing CommonInterface (contract of cleanmarl/env/smaclite_wrapper.py:12-60:
TimeLimit(150), real availability masks, obs (+ one-hot id), env-provided global state).  smaclite / gymnasium
are optional and imported lazily."""
import numpy as np

from .common_interface import CommonInterface


class SMACliteWrapper(CommonInterface):
    def __init__(self, map_name, seed=0, time_limit=150, agent_ids=False, **kwargs):
        try:
            import gymnasium as gym
            from gymnasium.wrappers import TimeLimit
            import smaclite  # noqa: F401  (registers the envs)
        except ImportError as e:  # pragma: no cover - optional dependency
            raise ImportError(f"env_type=smaclite needs 'smaclite' and 'gymnasium' ({e})") from e
        self.env = TimeLimit(gym.make(f"smaclite/{map_name}-v0", seed=seed, **kwargs), max_episode_steps=time_limit)
        self.agent_ids = bool(agent_ids)
        self.n_agents = self.env.unwrapped.n_agents
        self.episode_limit = time_limit
        self._n_act = max(int(s.n) for s in self.env.action_space)

    def _process(self, obs):
        obs = np.array(obs)
        return np.concatenate((obs, np.eye(self.n_agents)), axis=1) if self.agent_ids else obs

    def step(self, actions):
        obs, reward, terminated, truncated, info = self.env.step([int(a) for a in actions])
        return self._process(obs), reward, terminated, truncated, info

    def reset(self, seed=None, options=None):
        obs, _ = self.env.reset(seed=seed, options=options)
        return self._process(obs), {}

    def get_obs_size(self):
        return self.env.unwrapped.obs_size + self.agent_ids * self.n_agents

    def get_state_size(self):
        return self.env.unwrapped.state_size

    def get_state(self):
        return self.env.unwrapped.get_state()

    def get_action_size(self):
        return self._n_act

    def get_avail_actions(self):
        return np.array(self.env.unwrapped.get_avail_actions())

    def sample(self):
        av = np.asarray(self.get_avail_actions(), dtype=np.float64)
        return [int(np.random.choice(len(r), p=r / r.sum())) for r in av]

    def close(self):
        self.env.close()
