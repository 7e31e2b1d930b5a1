"""Device-resident rollout: on-device synthetic env + fused act kernel write straight into the
[E,A,T,F] rollout buffer (replaces cleanmarl/mappo_multienvs.py:393-453 + RolloutBuffer :82-157 for the
synthetic configs: no pipes, no per-step host round trip, no collate copy)."""
import ctypes as C

import torch

from . import _native as N
from .learner import DeviceBatch

_GOLD = 0x9E3779B97F4A7C15


def _off(t, nbytes):
    return C.c_void_p(t.data_ptr() + nbytes)


class SyntheticSpreadRollout:
    """E synthetic MPE-like envs (csrc/cm_env.hip), all truncated at exactly T steps."""

    def __init__(self, E, A, T, seed=1, agent_ids=True, device="cuda:0", env_offset=0, pad=True):
        """pad: obs / state buffers with leading dimensions rounded up to 4 floats (learner.DeviceBatch) -- the learner's kernels then
        read 16-byte aligned rows whatever the env's feature widths; pad=False for consumers of contiguous rows (COMA)."""
        self.lib = N.load()
        self.E, self.A, self.T, self.K = E, A, T, 5
        self.agent_ids = bool(agent_ids)
        self.Do = 6 * A + (A if agent_ids else 0)
        self.Ds = 6 * A * A
        self.seed, self.env_offset = int(seed), int(env_offset)
        self.device = torch.device(device)
        # two buffers used alternately: the learner's critic epochs (own stream, learner.py) may still read episode i's states and
        # returns while episode i + 1 is being written
        self.batches = [DeviceBatch(E, A, T, self.Do, self.Ds, self.K, self.device, pad_obs=pad, pad_state=pad) for _ in range(2)]
        for bb in self.batches:
            bb.avail.fill_(1)
            bb.ep_len.fill_(T)
        self.batch = self.batches[0]  # the most recently collected one
        self.env_state = torch.zeros(E, 6 * A, dtype=torch.float32, device=self.device)
        self.episode = 0

    def collect(self, actor_flat, actor_spec, fused=None, eps=0.0):
        """One episode per env (reference outer loop body, :393-453).  Everything is enqueued on the
        current stream; returns the filled DeviceBatch without synchronising.
        fused=None picks the single-launch persistent kernel (cm_rollout_spread) whenever the shape allows,
        else T x (cm_policy_act + cm_synth_env_step); both produce the same rollout for the same seeds."""
        N.sync_env_options()
        self.batch = self.batches[self.episode & 1]
        lib, b, s = self.lib, self.batch, N.stream_ptr()
        E, A, T, Do, K = self.E, self.A, self.T, self.Do, self.K
        can_fuse = actor_spec.kind == "mlp" and bool(lib.cm_rollout_spread_supported(A, int(self.agent_ids), actor_spec.hidden,
                                                                                 actor_spec.n_layers))
        if fused is None:
            fused = can_fuse
        if fused:
            if not can_fuse:
                raise N.NativeError("fused rollout requested for an unsupported shape")
            act_seed = (self.seed + (self.episode + 1) * _GOLD) & 0xFFFFFFFFFFFFFFFF
            # eps > 0: COMA's epsilon-mixed exploration (coma_multienvs.py:477-484)
            N.check(lib.cm_rollout_spread_ld(N.ptr(self.env_state), E, A, T, int(self.agent_ids), self.seed, act_seed, self.env_offset,
                                             self.episode, N.ptr(actor_flat), actor_spec.hidden, actor_spec.n_layers, float(eps),
                                             N.ptr(b.obs), b.obs_ld, N.ptr(b.state), b.state_ld, N.ptr(b.action), N.ptr(b.logp),
                                             N.ptr(b.reward), s), "cm_rollout_spread_ld")
            self.episode += 1
            return b
        if b.obs_ld != Do or b.state_ld != b.Ds:
            # the per-step env kernels (cm_synth_env_reset / _step) write contiguous rows: shapes the fused kernel does not cover (and
            # explicit fused=False runs) switch this rollout to unpadded buffers, once.  The learner's critic epochs may still read the
            # old buffers on their own stream: wait for the device before their storage goes back to the allocator (one-time cost)
            torch.cuda.synchronize(self.device)
            self.batches = [DeviceBatch(E, A, T, self.Do, self.Ds, self.K, self.device) for _ in range(2)]
            for bb in self.batches:
                bb.avail.fill_(1)
                bb.ep_len.fill_(T)
            self.batch = b = self.batches[self.episode & 1]
        N.check(lib.cm_synth_env_reset(N.ptr(self.env_state), E, A, int(self.agent_ids), self.seed, self.env_offset,
                                       self.episode, N.ptr(b.obs), N.ptr(b.state), T, s), "cm_synth_env_reset")
        act_seed = (self.seed + (self.episode + 1) * _GOLD) & 0xFFFFFFFFFFFFFFFF
        need = lib.cm_policy_act_workspace_bytes(E * A, actor_spec.din, actor_spec.hidden, actor_spec.n_layers, K)  # 0 unless layered
        if need and (getattr(self, "act_ws", None) is None or self.act_ws.numel() < need):
            self.act_ws = torch.empty(need, dtype=torch.uint8, device=self.device)
        ws = N.ptr(self.act_ws) if need else None
        for t in range(T):
            N.check(lib.cm_policy_act_ws(_off(b.obs, 4 * t * Do), T * Do, _off(b.avail, t * K), T * K, E * A,
                                         actor_spec.din, actor_spec.hidden, actor_spec.n_layers, K, N.ptr(actor_flat), float(eps),
                                         act_seed, self.env_offset * A, t, _off(b.action, 4 * t), _off(b.logp, 4 * t), T, ws, need, s),
                    "cm_policy_act_ws")
            N.check(lib.cm_synth_env_step(N.ptr(self.env_state), N.ptr(b.action), E, A, int(self.agent_ids), t, T,
                                          N.ptr(b.reward), N.ptr(b.obs), N.ptr(b.state), s), "cm_synth_env_step")
        self.episode += 1
        return b


class SyntheticShapeRollout:
    """Device rollout of the "shape" env (BASELINE config 4 shapes): one launch generates the episode's observations /
    states / availability masks, T x cm_policy_act (or cm_gru_policy_act) samples actions, one launch computes rewards."""

    def __init__(self, E, A, T, obs_raw=105, state_dim=243, n_actions=17, avail_p=0.7, seed=1, agent_ids=True, device="cuda:0",
                 env_offset=0, pad=True):
        self.lib = N.load()
        self.E, self.A, self.T, self.K = E, A, T, n_actions
        self.obs_raw, self.agent_ids, self.avail_p = obs_raw, bool(agent_ids), float(avail_p)
        self.Do, self.Ds = obs_raw + (A if agent_ids else 0), state_dim
        self.seed, self.env_offset, self.device = int(seed), int(env_offset), torch.device(device)
        self.batches = [DeviceBatch(E, A, T, self.Do, self.Ds, self.K, self.device, pad_obs=pad, pad_state=pad) for _ in range(2)]  # see SyntheticSpreadRollout
        for bb in self.batches:
            bb.ep_len.fill_(T)
        self.batch = self.batches[0]
        self.h = None
        self.episode = 0

    def collect(self, actor_flat, actor_spec, fused=None, eps=0.0):
        N.sync_env_options()
        self.batch = self.batches[self.episode & 1]
        lib, b, s = self.lib, self.batch, N.stream_ptr()
        E, A, T, Do, K = self.E, self.A, self.T, self.Do, self.K
        need = 0
        if actor_spec.kind != "gru":  # layered schedule (hidden > 64 / deeper than 2 hidden layers): per-step act with a workspace
            need = lib.cm_policy_act_workspace_bytes(E * A, actor_spec.din, actor_spec.hidden, actor_spec.n_layers, K)
            if need and (getattr(self, "act_ws", None) is None or self.act_ws.numel() < need):
                self.act_ws = torch.empty(need, dtype=torch.uint8, device=self.device)
        if eps != 0.0 or need:  # eps > 0: COMA's mixture; eps < 0: greedy evaluation rollouts -- the per-step act kernel has both samplers
            fused = False
        N.check(lib.cm_shape_env_fill_ld(E, A, T, self.obs_raw, int(self.agent_ids), self.Ds, K, self.avail_p, self.seed, self.env_offset,
                                         self.episode, N.ptr(b.obs), b.obs_ld, N.ptr(b.state), b.state_ld, N.ptr(b.avail), s),
                "cm_shape_env_fill_ld")
        act_seed = (self.seed + (self.episode + 1) * _GOLD) & 0xFFFFFFFFFFFFFFFF
        gru = actor_spec.kind == "gru"
        if not gru and fused is not False:  # obs do not depend on the actions: sample every step in ONE launch
            w0b = lib.cm_w0_image_bytes(actor_spec.din, actor_spec.hidden)  # scratch for the padded image of a streamed, unaligned W0
            if w0b and (getattr(self, "w0_ws", None) is None or self.w0_ws.numel() < w0b):
                self.w0_ws = torch.empty(w0b, dtype=torch.uint8, device=self.device)
            N.check(lib.cm_policy_act_episode_ld(N.ptr(b.obs), b.obs_ld, N.ptr(b.avail), E * A, T, actor_spec.din, actor_spec.hidden,
                                                 actor_spec.n_layers, K, N.ptr(actor_flat), act_seed, self.env_offset * A,
                                                 N.ptr(b.action), N.ptr(b.logp), N.ptr(self.w0_ws) if w0b else None, w0b, s),
                    "cm_policy_act_episode_ld")
            N.check(lib.cm_shape_env_reward(E, A, T, K, self.seed, self.env_offset, self.episode, N.ptr(b.action), N.ptr(b.reward), s),
                    "cm_shape_env_reward")
            self.episode += 1
            return b
        if gru:
            if self.h is None:
                self.h = torch.zeros(E * A, actor_spec.hidden, dtype=torch.float32, device=self.device)
            self.h.zero_()
            gneed = lib.cm_gru_policy_act_workspace_bytes(E * A, actor_spec.din, actor_spec.hidden, K)  # used by the layered shapes only
            if getattr(self, "gru_ws", None) is None or self.gru_ws.numel() < gneed:
                self.gru_ws = torch.empty(gneed, dtype=torch.uint8, device=self.device)
        for t in range(T):
            if gru:
                N.check(lib.cm_gru_policy_act_ws(_off(b.obs, 4 * t * b.obs_ld), T * b.obs_ld, _off(b.avail, t * K), T * K, E * A, actor_spec.din,
                                                 actor_spec.hidden, K, N.ptr(actor_flat), N.ptr(self.h), float(min(eps, 0.0)), act_seed, self.env_offset * A, t,
                                                 _off(b.action, 4 * t), _off(b.logp, 4 * t), T, N.ptr(self.gru_ws), self.gru_ws.numel(), s),
                        "cm_gru_policy_act_ws")
            else:
                N.check(lib.cm_policy_act_ws(_off(b.obs, 4 * t * b.obs_ld), T * b.obs_ld, _off(b.avail, t * K), T * K, E * A, actor_spec.din,
                                             actor_spec.hidden, actor_spec.n_layers, K, N.ptr(actor_flat), float(eps), act_seed,
                                             self.env_offset * A, t, _off(b.action, 4 * t), _off(b.logp, 4 * t), T,
                                             N.ptr(self.act_ws) if need else None, need, s), "cm_policy_act_ws")
        N.check(lib.cm_shape_env_reward(E, A, T, K, self.seed, self.env_offset, self.episode, N.ptr(b.action), N.ptr(b.reward), s),
                "cm_shape_env_reward")
        self.episode += 1
        return b
