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


def _normal01(a, b):
    """Box-Muller on two Philox words, fp32 -- twin of normal01() in csrc/cm_env.hip."""
    u1 = ((np.asarray(a, np.uint32) >> np.uint32(8)).astype(F) + F(0.5)) * F(1.0 / 16777216.0)
    u2 = u01(b)
    return (np.sqrt(F(-2.0) * np.log(u1)) * np.cos(F(6.283185307179586) * u2)).astype(F)


def _normal4(x, y, z, w):
    """FOUR normals per Philox call -- twin of normal4() in csrc/cm_env.hip: both Box-Muller outputs of the word pairs (x, y), (z, w).
    Returns [..., 4]."""
    def pair(a, b):
        u1 = ((np.asarray(a, np.uint32) >> np.uint32(8)).astype(F) + F(0.5)) * F(1.0 / 16777216.0)
        r = np.sqrt(F(-2.0) * np.log(u1)).astype(F)
        ang = (F(6.283185307179586) * u01(b)).astype(F)
        return (r * np.cos(ang)).astype(F), (r * np.sin(ang)).astype(F)
    n0, n1 = pair(x, y)
    n2, n3 = pair(z, w)
    return np.stack([n0, n1, n2, n3], axis=-1)


class SyntheticShapeEnv(CommonInterface):
    """CPU twin of the on-device "shape" env (cm_shape_env_fill / cm_shape_env_reward): wide random observations, a
    separate global state, availability masks with action 0 always legal, fixed horizon -- BASELINE config 4's
    SMAClite-like shapes (obs 105 + 10 ids, state 243, 17 actions) without the SMAClite dependency."""

    def __init__(self, n_agents=10, obs_raw=105, state_dim=243, n_actions=17, avail_p=0.7, agent_ids=True, max_cycles=256,
                 seed=1, env_index=0, **kwargs):
        self.n_agents, self.obs_raw, self.state_dim, self.K = int(n_agents), int(obs_raw), int(state_dim), int(n_actions)
        self.avail_p, self.agent_ids, self.max_cycles = F(avail_p), bool(agent_ids), int(max_cycles)
        self.seed, self.env_index = int(seed), int(env_index)
        self.episode, self.t = -1, 0

    def _words(self, c2, c3):
        k0, k1 = split_seed(self.seed)
        return philox4x32(np.uint32(self.env_index & 0xFFFFFFFF), np.uint32(self.episode), np.asarray(c2, np.uint32),
                          np.asarray(c3, np.uint32), k0, k1)

    def _observe(self):
        A, t = self.n_agents, self.t
        c2 = (t * A + np.arange(A, dtype=np.uint32))[:, None]
        # features / actions 4g .. 4g+3 share the Philox counter (.., base + g): four values per call (normal4 / the four words)
        go, gs, gk = (self.obs_raw + 3) // 4, (self.state_dim + 3) // 4, (self.K + 3) // 4
        raw = _normal4(*self._words(c2, np.uint32(0x100) + np.arange(go, dtype=np.uint32)[None, :])).reshape(A, 4 * go)[:, :self.obs_raw]
        self.state = _normal4(*self._words(np.uint32(t), np.uint32(0x40000000) + np.arange(gs, dtype=np.uint32))).reshape(4 * gs)[:self.state_dim]
        words = np.stack(self._words(c2, np.uint32(0x80000000) + np.arange(gk, dtype=np.uint32)[None, :]), axis=-1).reshape(A, 4 * gk)[:, :self.K]
        av = (u01(words) < self.avail_p)
        av[:, 0] = True
        self.avail = av.astype(np.int64)
        return np.concatenate([raw, np.eye(A, dtype=F)], 1) if self.agent_ids else raw

    def reset(self, seed=None):
        self.episode += 1
        self.t = 0
        return self._observe(), {}

    def step(self, actions):
        x, y, _, _ = self._words(np.uint32(self.t), np.uint32(0xC0000000))
        hits = sum(int(a) == self.t % self.K for a in actions)
        r = F(_normal01(x, y) + F(hits) / F(self.n_agents))
        self.t += 1
        return self._observe(), float(r), False, bool(self.t >= self.max_cycles), {}

    def get_avail_actions(self):
        return self.avail

    def get_action_size(self):
        return self.K

    def get_state(self):
        return self.state

    def get_state_size(self):
        return self.state_dim

    def get_obs_size(self):
        return self.obs_raw + self.agent_ids * self.n_agents

    def sample(self):
        return [int(np.random.choice(np.flatnonzero(r))) for r in self.avail]

    def close(self):
        pass
