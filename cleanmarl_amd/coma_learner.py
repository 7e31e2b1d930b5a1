"""COMA learner on the HIP kernels: the update of cleanmarl/coma_multienvs.py:553-684 (identical in
cleanmarl/coma.py) on a DeviceBatch -- targets from the TARGET critic (TD(lambda) via cm_td_lambda_scan with
V := Q_target[taken action], or cm_nstep_returns), ONE critic step (cm_coma_critic_fwd_bwd: factored critic input, never materialised), polyak target update, ONE actor
step with the counterfactual baseline (cm_coma_advantage / cm_coma_normalize_adv / cm_coma_actor_fwd_bwd).

Env-sharded data parallelism as in learner.py: per optimiser step one all-reduce(sum) of the un-normalised flat gradient
+ statistics buffer; the per-time-step advantage moments are all-reduced as raw float64 sums [T][4].
"""
from collections.abc import Mapping
from dataclasses import dataclass

import torch

from . import _native as N
from . import dist
from .learner import LazyRecords, NetSpec, _Adam, _HostRing, _to_host_async, flatten_params, init_params_like_torch


@dataclass
class COMAHParams:
    """Learner-relevant subset of cleanmarl/coma_multienvs.py:19-89."""
    gamma: float = 0.99
    td_lambda: float = 0.8
    normalize_reward: bool = False
    normalize_advantage: bool = True
    normalize_return: bool = False
    target_network_update_freq: int = 1
    polyak: float = 0.005
    entropy_coef: float = 0.001
    use_tdlamda: bool = True
    nsteps: int = 1
    clip_gradients: float = -1
    optimizer: str = "Adam"
    learning_rate_actor: float = 0.0005
    learning_rate_critic: float = 0.0005

    @classmethod
    def from_args(cls, args):
        return cls(**{f: getattr(args, f) for f in cls.__dataclass_fields__ if hasattr(args, f)})


def coma_critic_input_dim(Do, Ds, A, K):
    """get_coma_critic_input_dim, cleanmarl/coma_multienvs.py:273-279."""
    return Do + Ds + (A - 1) * K


class LazyRecord(Mapping):
    """The record of one COMA iteration as a read-only mapping whose numbers are still on their way from the device (same staging
    as learner.LazyRecords: pinned ring buffer, converted on first access) -- a caller that does not log every iteration no longer
    stalls the launch queue once per iteration (kernel trace of tools/bench_coma.py: 131 us before the next rollout)."""

    def __init__(self, lazy):
        self._lazy = lazy

    def _d(self):
        return self._lazy[0]

    def __getitem__(self, k):
        return self._d()[k]

    def __iter__(self):
        return iter(self._d())

    def __len__(self):
        return len(self._d())

    def __reduce__(self):  # pickles (torch.save) as a plain dict
        return (dict, (dict(self._d()),))


class COMALearner:
    def __init__(self, actor_spec, critic_spec, n_agents, hp, device, actor_params=None, critic_params=None,
                 process_group=None, world_size=1):
        self.lib = N.load()
        self.A, self.hp, self.device = n_agents, hp, device
        self.actor_spec, self.critic_spec = actor_spec, critic_spec
        assert critic_spec.dout == actor_spec.dout, "COMA's critic has one output per action"
        self.pg, self.world = process_group, world_size
        # CM_FORCE_COLLECTIVES=1 (test hook, as in learner.PPOLearner): issue the all-reduces at world size 1 too
        import os
        self._coll = world_size > 1 or (process_group is not None and os.environ.get("CM_FORCE_COLLECTIVES") == "1")
        self.actor = flatten_params(actor_params if actor_params is not None else init_params_like_torch(actor_spec), device)
        self.critic = flatten_params(critic_params if critic_params is not None else init_params_like_torch(critic_spec), device)
        self.target = self.critic.clone()  # copy.deepcopy(critic), :406
        Pa, Pc = actor_spec.nparams, critic_spec.nparams
        self.opt_a = _Adam(Pa, hp.learning_rate_actor, hp.optimizer, device)
        self.opt_c = _Adam(Pc, hp.learning_rate_critic, hp.optimizer, device)
        self.g_actor = torch.zeros(Pa + N.NUM_STATS, dtype=torch.float32, device=device)
        self.g_critic = torch.zeros(Pc + N.NUM_STATS, dtype=torch.float32, device=device)
        self.norms = torch.zeros(2, dtype=torch.float32, device=device)
        self.moments = torch.zeros(3, dtype=torch.float64, device=device)
        self.training_step = 0
        self._shape = None
        self.events = None
        self._ring = _HostRing()

    # ------------------------------------------------------------------ buffers sized on first use
    def _ensure(self, b):
        shape = (b.E, b.A, b.T)
        if self._shape == shape:
            return
        E, A, T, K = b.E, b.A, b.T, b.K
        cs, a, dev, lib = self.critic_spec, self.actor_spec, self.device, self.lib
        f32 = dict(dtype=torch.float32, device=dev)
        assert cs.din == coma_critic_input_dim(b.Do, b.Ds, A, K), (cs.din, b.Do, b.Ds, A, K)
        self.q = torch.empty(E, A, T, K, **f32)
        self.logits = torch.empty(E, A, T, K, **f32)
        self.qtaken = torch.empty(E, A, T, **f32)
        self.scratch = torch.empty(E, A, T, **f32)            # the scan's unused advantage output
        self.tstats = torch.zeros(T, 4, dtype=torch.float64, device=dev)
        rows = E * A * T
        need = max(lib.cm_coma_critic_workspace_bytes(E, A, T, b.Ds, b.Do, K, cs.hidden, cs.n_layers, 1),
                   lib.cm_mlp_split_workspace_bytes(rows, a.din, a.hidden, a.n_layers, K),
                   lib.cm_mlp_forward_workspace_bytes(rows, a.din, a.hidden, a.n_layers, K),
                   lib.cm_coma_advantage_workspace_bytes(E, A, T), lib.cm_masked_moments_workspace_bytes(E, A, T))
        self.ws = torch.empty(need, dtype=torch.uint8, device=dev)
        self._shape = shape

    def _moments(self, x, ep_len, E, A, T, s):
        if E == 0:
            self.moments.zero_()
        else:
            N.check(self.lib.cm_masked_moments(N.ptr(x), N.ptr(ep_len), E, A, T, N.ptr(self.moments), N.ptr(self.ws), self.ws.numel(), s),
                    "cm_masked_moments")
        dist.merge_moments_(self.moments, self.pg, self.world)

    def _empty_shard(self, b):
        """A rank of an env-sharded run that owns no environments (fewer envs than ranks): no kernel is launched, its [gradient |
        statistics] buffers, moment triples and per-time-step advantage sums are zero, every collective and optimiser step still happens
        (same contract as learner.PPOLearner._empty_shard)."""
        if b.E > 0:
            return False
        if not self._coll or self.world <= 1:
            raise N.NativeError("empty batch (0 environments): only a rank of an env-sharded run (world size > 1) may own no environments")
        return True

    def _adam(self, params, g, opt, which, s):
        """norm_d + clip_grad_norm_ + optimizer.step() as one launch on the (all-reduced) gradient buffer."""
        o = opt.next_step(params, self.norms[which:], self.hp.clip_gradients)
        N.check(self.lib.cm_optimizer_step(N.ptr(g), params.numel(), o, s), "cm_optimizer_step")

    def _allreduce(self, t):
        if self._coll:
            torch.distributed.all_reduce(t, group=self.pg)

    def _q(self, params, avail, out, b, s):
        """Q[E,A,T,K] of Critic(state, obs, actions) without materialising coma_inputs (factored layer 0, csrc/cm_coma.hip)."""
        cs = self.critic_spec
        N.check(self.lib.cm_coma_q_forward_ld(N.ptr(b.state), b.state_ld, N.ptr(b.obs), b.obs_ld, N.ptr(b.action),
                                              N.ptr(avail) if avail is not None else None, b.E, b.A, b.T, b.Ds, b.Do, b.K, cs.hidden, cs.n_layers,
                                              N.ptr(params), N.ptr(out), N.ptr(self.ws), self.ws.numel(), s), "cm_coma_q_forward_ld")

    # ------------------------------------------------------------------ :553-618
    def compute_targets(self, b):
        N.sync_env_options()
        lib, hp, s = self.lib, self.hp, N.stream_ptr()
        E, A, T, K = b.E, b.A, b.T, b.K
        if self._empty_shard(b):
            for on, Am in ((hp.normalize_reward, 1), (hp.normalize_return, A)):
                if on:
                    self._moments(None, None, 0, Am, T, s)
            return
        self._ensure(b)
        if hp.normalize_reward:  # RolloutBuffer.get_batch, :151-154
            self._moments(b.reward, b.ep_len, E, 1, T, s)
            N.check(lib.cm_normalize(N.ptr(b.reward), N.ptr(b.ep_len), E, 1, T, N.ptr(self.moments), 1e-6, 1, s), "cm_normalize")
        self._q(self.target, b.avail, self.q, b, s)                                 # target critic, masked (:565-570)
        N.check(lib.cm_gather_taken(N.ptr(self.q), N.ptr(b.action), E * A * T, K, N.ptr(self.qtaken), s), "cm_gather_taken")
        if hp.use_tdlamda:
            N.check(lib.cm_td_lambda_scan(N.ptr(b.reward), N.ptr(self.qtaken), N.ptr(b.ep_len), E, A, A, T, hp.gamma, hp.td_lambda,
                                          N.ptr(b.ret), N.ptr(self.scratch), s), "cm_td_lambda_scan")
        else:
            N.check(lib.cm_nstep_returns(N.ptr(b.reward), N.ptr(self.qtaken), N.ptr(b.ep_len), E, A, T, hp.gamma, int(hp.nsteps),
                                         N.ptr(b.ret), s), "cm_nstep_returns")
        if hp.normalize_return:  # :615-618
            self._moments(b.ret, b.ep_len, E, A, T, s)
            N.check(lib.cm_normalize(N.ptr(b.ret), N.ptr(b.ep_len), E, A, T, N.ptr(self.moments), 0.0, 0, s), "cm_normalize")

    # ------------------------------------------------------------------ :620-684
    def update(self, b, keep_grads=False):
        N.sync_env_options()
        lib, hp, s = self.lib, self.hp, N.stream_ptr()
        E, A, T, K = b.E, b.A, b.T, b.K
        cs, a = self.critic_spec, self.actor_spec
        Pa, Pc = self.actor.numel(), self.critic.numel()
        if self._empty_shard(b):
            return self._update_empty(b, keep_grads)
        self._ensure(b)
        # ---- critic step
        N.check(lib.cm_coma_critic_fwd_bwd_ld(N.ptr(b.state), b.state_ld, N.ptr(b.obs), b.obs_ld, N.ptr(b.action), N.ptr(b.ret), N.ptr(b.ep_len),
                                              E, A, T, b.Ds, b.Do, K, cs.hidden, cs.n_layers, N.ptr(self.critic), N.ptr(self.g_critic),
                                              N.ptr(self.ws), self.ws.numel(), s), "cm_coma_critic_fwd_bwd_ld")
        self._allreduce(self.g_critic)
        self._adam(self.critic, self.g_critic, self.opt_c, 1, s)
        self.training_step += 1
        if self.training_step % int(hp.target_network_update_freq) == 0:
            N.check(lib.cm_polyak_update(N.ptr(self.target), N.ptr(self.critic), Pc, hp.polyak, s), "cm_polyak_update")
        # ---- actor step: Q of the UPDATED critic (no availability mask, :655-657), counterfactual advantage
        self._q(self.critic, None, self.q, b, s)
        N.check(lib.cm_mlp_forward_ld(N.ptr(b.obs), b.obs_ld, E * A * T, a.din, a.hidden, a.n_layers, K, N.ptr(self.actor), N.ptr(b.avail),
                                      N.ptr(self.logits), N.ptr(self.ws), self.ws.numel(), s), "cm_mlp_forward_ld")
        N.check(lib.cm_coma_advantage(N.ptr(self.logits), N.ptr(self.q), N.ptr(b.action), N.ptr(b.ep_len), E, A, T, K, N.ptr(b.adv),
                                      N.ptr(self.tstats), N.ptr(self.ws), self.ws.numel(), s), "cm_coma_advantage")
        if hp.normalize_advantage:
            self._allreduce(self.tstats)
            N.check(lib.cm_coma_normalize_adv(N.ptr(b.adv), N.ptr(self.tstats), E, A, T, s), "cm_coma_normalize_adv")
        N.check(lib.cm_coma_actor_fwd_bwd_ld(N.ptr(b.obs), b.obs_ld, N.ptr(b.avail), N.ptr(b.action), N.ptr(b.adv), N.ptr(b.ep_len), E, A, T,
                                             a.din, a.hidden, a.n_layers, K, N.ptr(self.actor), hp.entropy_coef, N.ptr(self.g_actor),
                                             N.ptr(self.ws), self.ws.numel(), s), "cm_coma_actor_fwd_bwd_ld")
        self._allreduce(self.g_actor)
        self._adam(self.actor, self.g_actor, self.opt_a, 0, s)
        return self._records(keep_grads)

    def _update_empty(self, b, keep_grads):
        """update() of a rank without environments: the same collectives and steps on all-zero contributions."""
        lib, hp, s = self.lib, self.hp, N.stream_ptr()
        self.g_critic.zero_()
        self._allreduce(self.g_critic)
        self._adam(self.critic, self.g_critic, self.opt_c, 1, s)
        self.training_step += 1
        if self.training_step % int(hp.target_network_update_freq) == 0:
            N.check(lib.cm_polyak_update(N.ptr(self.target), N.ptr(self.critic), self.critic.numel(), hp.polyak, s), "cm_polyak_update")
        if hp.normalize_advantage:
            if getattr(self, "tstats", None) is None or self.tstats.shape[0] != b.T:
                self.tstats = torch.zeros(b.T, 4, dtype=torch.float64, device=self.device)
            self.tstats.zero_()
            self._allreduce(self.tstats)
        self.g_actor.zero_()
        self._allreduce(self.g_actor)
        self._adam(self.actor, self.g_actor, self.opt_a, 0, s)
        return self._records(keep_grads)

    def _records(self, keep_grads):
        hp = self.hp
        Pa, Pc = self.actor.numel(), self.critic.numel()
        ent_coef, tstep = hp.entropy_coef, self.training_step
        extra = dict(actor_grads=self.g_actor[:Pa].clone(), critic_grads=self.g_critic[:Pc].clone()) if keep_grads else {}

        def build(st):
            st_a, st_c = st[:N.NUM_STATS], st[N.NUM_STATS:2 * N.NUM_STATS]
            n = float(st_a[N.STAT_COUNT])
            rec = dict(actor_loss=float(-st_a[N.STAT_PG] - ent_coef * st_a[N.STAT_ENT]) / n,
                       critic_loss=float(st_c[N.STAT_VLOSS]) / float(st_c[N.STAT_COUNT]), entropy=float(st_a[N.STAT_ENT]) / n,
                       actor_gnorm=float(st[2 * N.NUM_STATS]), critic_gnorm=float(st[2 * N.NUM_STATS + 1]), n_valid=n,
                       training_step=tstep)
            rec.update(extra)
            return [rec]
        # no host wait here (learner.LazyRecords): the statistics go to a pinned ring slot asynchronously
        host, ev, attach = _to_host_async(self._ring, torch.cat([self.g_actor[Pa:], self.g_critic[Pc:], self.norms]))
        lazy = LazyRecords(1, host, ev, build)
        attach(lazy)
        return LazyRecord(lazy)

    def critic_params(self):
        """Same accessor as PPOLearner.critic_params (COMA's critic runs on the launch stream: nothing to join)."""
        return self.critic

    def train_iteration(self, b, keep_grads=False):
        self.compute_targets(b)
        return self.update(b, keep_grads=keep_grads)

    # ------------------------------------------------------------------ checkpointing
    def state_dict(self):
        return dict(algo="coma", actor_spec=vars(self.actor_spec), critic_spec=vars(self.critic_spec), actor=self.actor.cpu(),
                    critic=self.critic.cpu(), target=self.target.cpu(), training_step=self.training_step,
                    opt_a=dict(m=self.opt_a.m.cpu(), v=self.opt_a.v.cpu(), step=self.opt_a.step),
                    opt_c=dict(m=self.opt_c.m.cpu(), v=self.opt_c.v.cpu(), step=self.opt_c.step))

    def load_state_dict(self, sd):
        if sd.get("algo") != "coma" or sd["actor_spec"] != vars(self.actor_spec) or sd["critic_spec"] != vars(self.critic_spec):
            raise N.NativeError("checkpoint was written for a different algorithm / network shape")
        self.actor.copy_(sd["actor"]); self.critic.copy_(sd["critic"]); self.target.copy_(sd["target"])
        self.training_step = int(sd["training_step"])
        for opt, o in ((self.opt_a, sd["opt_a"]), (self.opt_c, sd["opt_c"])):
            opt.m.copy_(o["m"]); opt.v.copy_(o["v"]); opt.step = int(o["step"])


__all__ = ["COMAHParams", "COMALearner", "NetSpec", "coma_critic_input_dim"]
