"""PettingZoo parallel-env adapter implementing CommonInterface.  Behavioural contract follows
cleanmarl/env/pettingzoo_wrapper.py:9-101: per-agent flattened obs (+ one-hot id), state = concat of raw obs
(:93-98), team reward = FIRST agent's reward (:66), all-ones availability padded to the longest action space
(:79-90), last obs re-used when the episode terminates (:58-64).  pettingzoo / gymnasium are imported lazily:
they are optional and absent from the build image."""
import importlib

import numpy as np

from .common_interface import CommonInterface


class PettingZooWrapper(CommonInterface):
    def __init__(self, family, env_name, agent_ids=False, **kwargs):
        try:
            mod = importlib.import_module(f"pettingzoo.{family}.{env_name}")
        except ImportError as e:  # pragma: no cover - optional dependency
            raise ImportError(f"env_type=pz needs the 'pettingzoo' package ({e}); use --env_type=synthetic_cpu for the "
                              "bundled MPE-like env") from e
        self.env = mod.parallel_env(**kwargs)
        self.env.reset()
        self.agents = list(self.env.agents)
        self.n_agents = len(self.agents)
        self.agent_ids = bool(agent_ids)
        self._obs_dim = max(int(np.prod(self.env.observation_space(a).shape)) for a in self.agents)
        self._n_act = [int(self.env.action_space(a).n) for a in self.agents]
        self._max_act = max(self._n_act)
        self.last_obs = None

    def _process(self, obs):
        raw = np.array([np.asarray(obs[a]).flatten() for a in self.agents])
        self.state = raw.reshape(-1)
        return np.concatenate((raw, np.eye(self.n_agents)), axis=1) if self.agent_ids else raw

    def reset(self, seed=None):
        obs, _ = self.env.reset(seed=seed)
        self.last_obs = self._process(obs)
        return self.last_obs, {}

    def step(self, actions):
        acts = {a: int(actions[i]) for i, a in enumerate(self.agents)}
        obs, rewards, dones, truncs, infos = self.env.step(acts)
        done = all(dones[a] for a in self.agents) if dones else True
        truncated = all(truncs[a] for a in self.agents) if truncs else False
        if done and not obs:
            out, rew = self.last_obs, [0.0] * self.n_agents
        else:
            out, rew = self._process(obs), [rewards[a] for a in self.agents]
            self.last_obs = out
        info = {f"{a}_{k}": v for a in self.agents for k, v in infos.get(a, {}).items()}
        return out, rew[0], done, truncated, info

    def get_obs_size(self):
        return self._obs_dim + self.agent_ids * self.n_agents

    def get_state_size(self):
        return self._obs_dim * self.n_agents

    def get_state(self):
        return self.state

    def get_action_size(self):
        return self._n_act[0]

    def get_avail_actions(self):
        return np.array([[1] * n + [0] * (self._max_act - n) for n in self._n_act])

    def sample(self):
        return [self.env.action_space(a).sample() for a in self.agents]

    def close(self):
        return self.env.close()
