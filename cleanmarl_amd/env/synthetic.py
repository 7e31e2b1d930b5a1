"""CPU (numpy, fp32) twin of the on-device synthetic MPE-like env in csrc/cm_env.hip.

Implements CommonInterface so it can sit behind the reference's env_worker pipe protocol
(cleanmarl/mappo_multienvs.py:246-285) -- used for the config-1 plumbing path, for the CPU baseline and as
the independent implementation the HIP env kernels are parity-tested against.
Shapes follow MPE simple_spread as wrapped by cleanmarl/env/pettingzoo_wrapper.py:68-73, 93-98:
Do = 6A (+A one-hot ids), Ds = 6A*A, 5 actions, all always available.
"""
import numpy as np

from .common_interface import CommonInterface
from .philox import STREAM_ENV_RESET, philox4x32, split_seed, u01

F = np.float32
DAMP, DT, ACCEL, COLLIDE = F(0.25), F(0.1), F(5.0), F(0.3)


class SyntheticSpreadEnv(CommonInterface):
    def __init__(self, n_agents=3, agent_ids=True, max_cycles=25, seed=1, env_index=0, **kwargs):
        self.n_agents = int(n_agents)
        self.agent_ids = bool(agent_ids)
        self.max_cycles = int(max_cycles)
        self.seed = int(seed)
        self.env_index = int(env_index)
        self.episode = -1
        self.t = 0

    # ---- CommonInterface
    def reset(self, seed=None):
        """`seed` is accepted for API compatibility; episodes are keyed by (self.seed, env_index, episode)
        so that runs are reproducible (the reference's worker seeds are not: SURVEY.md §8b)."""
        A = self.n_agents
        self.episode += 1
        self.t = 0
        k0, k1 = split_seed(self.seed)
        x, y, z, w = philox4x32(np.full(A, self.env_index & 0xFFFFFFFF, np.uint32), np.uint32(self.episode),
                                np.arange(A, dtype=np.uint32), np.uint32(STREAM_ENV_RESET), k0, k1)
        self.pos = np.stack([F(2) * u01(x) - F(1), F(2) * u01(y) - F(1)], 1).astype(F)
        self.lm = np.stack([F(2) * u01(z) - F(1), F(2) * u01(w) - F(1)], 1).astype(F)
        self.vel = np.zeros((A, 2), F)
        return self._obs(), {}

    def step(self, actions):
        a = np.asarray([int(k) for k in actions])
        u = np.zeros((self.n_agents, 2), F)
        u[a == 1, 0] = -ACCEL; u[a == 2, 0] = ACCEL; u[a == 3, 1] = -ACCEL; u[a == 4, 1] = ACCEL
        self.vel = (self.vel * (F(1) - DAMP) + u * DT).astype(F)
        self.pos = (self.pos + self.vel * DT).astype(F)
        d = np.sqrt(((self.pos[None, :, :] - self.lm[:, None, :]) ** 2).sum(-1, dtype=F)).astype(F)  # [landmark, agent]
        r = F(0)
        for l in range(self.n_agents):
            r = F(r - d[l].min())
        dd = np.sqrt(((self.pos[:, None, :] - self.pos[None, :, :]) ** 2).sum(-1, dtype=F))
        iu = np.triu_indices(self.n_agents, 1)
        r = F(r - F((dd[iu] < COLLIDE).sum()))
        self.t += 1
        truncated = self.t >= self.max_cycles
        return self._obs(), float(r), False, bool(truncated), {}

    def get_avail_actions(self):
        return np.ones((self.n_agents, 5), dtype=np.int64)

    def get_action_size(self):
        return 5

    def get_state(self):
        return self.state

    def get_state_size(self):
        return 6 * self.n_agents * self.n_agents

    def get_obs_size(self):
        return 6 * self.n_agents + self.agent_ids * self.n_agents

    def sample(self):
        return list(np.random.randint(0, 5, self.n_agents))

    def close(self):
        pass

    # ---- helpers
    def _obs(self):
        A = self.n_agents
        rows = []
        for i in range(A):
            others = [self.pos[j] - self.pos[i] for j in range(A) if j != i]
            rows.append(np.concatenate([self.vel[i], self.pos[i], (self.lm - self.pos[i]).reshape(-1)] + others +
                                       [np.zeros(2 * (A - 1), F)]).astype(F))
        raw = np.stack(rows)
        self.state = raw.reshape(-1).copy()
        if self.agent_ids:
            raw = np.concatenate([raw, np.eye(A, dtype=F)], 1)
        return raw
