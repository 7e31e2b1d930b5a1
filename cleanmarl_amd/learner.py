"""Host side of the MAPPO / IPPO learner: owns device buffers and sequences the HIP kernels.

Mirrors the inline learner of the reference script (cleanmarl/mappo_multienvs.py:481-612):
    TD(lambda) targets (:484-504) -> optional normalisations (:505-512) -> `epochs` full-batch PPO
    steps (:521-594) -> logged scalars (:597-612)
but every numeric step is a call into libcleanmarl_hip.so on the current HIP stream.  PyTorch only
provides device memory, streams and (for env-sharded multi-GPU runs) torch.distributed.
"""
import os
from dataclasses import dataclass

import torch

from . import _native as N
from . import dist


@dataclass
class NetSpec:
    """Shape of one MLP (cleanmarl/mappo_multienvs.py:160-200) or GRU actor (mappo_lstm_multienvs.py:162-168)."""
    din: int
    hidden: int
    n_layers: int
    dout: int
    kind: str = "mlp"

    @property
    def nparams(self):
        lib = N.load()
        if self.kind == "gru":
            return int(lib.cm_gru_param_count(self.din, self.hidden, self.dout))
        return int(lib.cm_mlp_param_count(self.din, self.hidden, self.n_layers, self.dout))

    def shapes(self):
        """Parameter tensor shapes in torch ``parameters()`` order."""
        H, D, K = self.hidden, self.din, self.dout
        if self.kind == "gru":
            return [(H, D), (H,), (3 * H, H), (3 * H, H), (3 * H,), (3 * H,), (K, H), (K,)]
        s = [(H, D), (H,)]
        for _ in range(self.n_layers):
            s += [(H, H), (H,)]
        return s + [(K, H), (K,)]


def init_params_like_torch(spec):
    """nn.Linear / nn.GRUCell default init (U(+-1/sqrt(fan_in))), drawn on the CPU in the reference's
    construction order so `torch.manual_seed(seed)` reproduces the reference's initial weights
    (cleanmarl/mappo_multienvs.py:329-339)."""
    import torch.nn as nn
    mods = []
    if spec.kind == "gru":
        mods = [nn.Linear(spec.din, spec.hidden), nn.GRUCell(spec.hidden, spec.hidden), nn.Linear(spec.hidden, spec.dout)]
    else:
        mods = [nn.Linear(spec.din, spec.hidden)] + [nn.Linear(spec.hidden, spec.hidden) for _ in range(spec.n_layers)]
        mods.append(nn.Linear(spec.hidden, spec.dout))
    return [p.detach().clone() for m in mods for p in m.parameters()]


def flatten_params(plist, device):
    return torch.cat([p.reshape(-1).float() for p in plist]).contiguous().to(device)


class DeviceBatch:
    """Rollout storage in the device layout of include/cleanmarl_hip.h (the [E,A,T,F] permutation of the
    reference's RolloutBuffer batch, cleanmarl/mappo_multienvs.py:109-157)."""

    def __init__(self, E, A, T, Do, Ds, K, device, pad_obs=False, pad_state=False):
        """pad_obs / pad_state: allocate the feature axis with a leading dimension rounded up to 4 floats (zero padding), so that rows are
        16-byte aligned for the MLP kernels' tile loads when Do / Ds is not a multiple of 4 (21, 35, 115 at the BASELINE configs).
        `obs` / `state` are then strided VIEWS [.., :Do] of the padded storage; `obs_ld` / `state_ld` are the leading dimensions that the
        `*_ld` entry points take.  Consumers that assume contiguous rows (GRU, COMA) get unpadded batches."""
        self.E, self.A, self.T, self.Do, self.Ds, self.K = E, A, T, Do, Ds, K
        self.device = device
        f32 = dict(dtype=torch.float32, device=device)
        self.obs_ld = (Do + 3) // 4 * 4 if pad_obs else Do
        self.state_ld = (Ds + 3) // 4 * 4 if pad_state else Ds
        self.obs = torch.zeros(E, A, T, self.obs_ld, **f32)[..., :Do]
        self.state = torch.zeros(E, T, self.state_ld, **f32)[..., :Ds]
        self.avail = torch.zeros(E, A, T, K, dtype=torch.uint8, device=device)
        self.action = torch.zeros(E, A, T, dtype=torch.int32, device=device)
        self.logp = torch.zeros(E, A, T, **f32)
        self.reward = torch.zeros(E, T, **f32)
        self.ep_len = torch.zeros(E, dtype=torch.int32, device=device)
        self.ret = torch.zeros(E, A, T, **f32)
        self.adv = torch.zeros(E, A, T, **f32)

    def shard(self, lo, hi):
        """View of the envs [lo, hi): what a rank of an env-sharded run owns (no copy)."""
        s = DeviceBatch.__new__(DeviceBatch)
        s.E, s.A, s.T, s.Do, s.Ds, s.K, s.device = hi - lo, self.A, self.T, self.Do, self.Ds, self.K, self.device
        s.obs_ld, s.state_ld = self.obs_ld, self.state_ld
        for k in ("obs", "state", "avail", "action", "logp", "reward", "ep_len", "ret", "adv"):
            setattr(s, k, getattr(self, k)[lo:hi])
        return s

    @classmethod
    def from_reference_layout(cls, b_obs, b_actions, b_log_probs, b_reward, b_states, b_avail, b_mask, device, pad=False):
        """Build from tensors laid out like the reference batch ([B,T,A,F] etc.).  pad: round the leading dimensions of obs / state up to 4."""
        B, T, A, Do = b_obs.shape
        self = cls(B, A, T, Do, b_states.shape[-1], b_avail.shape[-1], device, pad_obs=pad, pad_state=pad)
        self.obs.copy_(b_obs.permute(0, 2, 1, 3))
        self.state.copy_(b_states)
        self.avail.copy_(b_avail.permute(0, 2, 1, 3).to(torch.uint8))
        self.action.copy_(b_actions.permute(0, 2, 1).to(torch.int32))
        self.logp.copy_(b_log_probs.permute(0, 2, 1))
        self.reward.copy_(b_reward)
        self.ep_len.copy_(b_mask.sum(1).to(torch.int32))
        return self


def pad_time(b, T):
    """Zero-pad a DeviceBatch along the time axis to T steps (ep_len unchanged: the extra steps are masked).  Used so
    that env-sharded ranks with host envs agree on T (same TBPTT chunk schedule => same number of all-reduces)."""
    if T == b.T:
        return b
    assert T > b.T
    n = DeviceBatch(b.E, b.A, T, b.Do, b.Ds, b.K, b.device, pad_obs=b.obs_ld != b.Do, pad_state=b.state_ld != b.Ds)
    n.obs[:, :, :b.T] = b.obs; n.state[:, :b.T] = b.state; n.avail[:, :, :b.T] = b.avail; n.action[:, :, :b.T] = b.action
    n.logp[:, :, :b.T] = b.logp; n.reward[:, :b.T] = b.reward; n.ep_len.copy_(b.ep_len)
    return n


@dataclass
class HParams:
    """The learner-relevant subset of the reference's Args (cleanmarl/mappo_multienvs.py:18-79)."""
    gamma: float = 0.99
    td_lambda: float = 0.95
    normalize_reward: bool = False
    normalize_advantage: bool = False
    normalize_return: bool = False
    epochs: int = 3
    ppo_clip: float = 0.2
    entropy_coef: float = 0.001
    clip_gradients: float = -1
    optimizer: str = "Adam"
    learning_rate_actor: float = 0.0008
    learning_rate_critic: float = 0.0008
    tbptt: int = 10

    @classmethod
    def from_args(cls, args):
        return cls(**{f: getattr(args, f) for f in cls.__dataclass_fields__ if hasattr(args, f)})


class _Adam:
    """Optimiser state for one flat parameter buffer.  The reference looks the class up with getattr(optim, args.optimizer) and
    passes only lr (cleanmarl/mappo_multienvs.py:341-343), so each kind runs with torch's defaults: Adam / AdamW (wd 0.01) /
    SGD (no momentum) / RMSprop (alpha 0.99, eps 1e-8)."""
    KINDS = {"Adam": N.OPT_ADAM, "AdamW": N.OPT_ADAMW, "SGD": N.OPT_SGD, "RMSprop": N.OPT_RMSPROP}

    def __init__(self, nparams, lr, kind, device):
        if kind not in self.KINDS:
            raise N.NativeError(f"optimizer={kind!r}: only {sorted(self.KINDS)} have a HIP implementation")
        self.kind = self.KINDS[kind]
        self.wd = 0.01 if kind == "AdamW" else 0.0
        self.beta2 = 0.99 if kind == "RMSprop" else 0.999
        self.lr = lr
        self.m = torch.zeros(nparams, dtype=torch.float32, device=device)
        self.v = torch.zeros(nparams, dtype=torch.float32, device=device)
        self.step = 0
        # ticket + per-workgroup sums of squares of the fused step launch (cm_opt_step_t.scratch): zeroed once, left zeroed by the kernels
        self.scratch = torch.zeros(N.load().cm_opt_step_scratch_bytes(), dtype=torch.uint8, device=device)

    def next_step(self, params, out_norm, max_norm, grad_scale=1.0, stats_out=None):
        """cm_opt_step_t of the NEXT optimiser step on `params` (advances the step counter).  stats_out: [8] device view that also receives
        the statistic sums (a row of the caller's record buffer: no copy launch afterwards)."""
        self.step += 1
        return N.OptStep(params=params.data_ptr(), exp_avg=self.m.data_ptr(), exp_avg_sq=self.v.data_ptr(), out_norm=out_norm.data_ptr(),
                         scratch=self.scratch.data_ptr(), lr=self.lr, beta1=0.9, beta2=self.beta2, eps=1e-8, weight_decay=self.wd,
                         max_norm=float(max_norm), grad_scale=float(grad_scale), step=self.step, opt_kind=self.kind,
                         stats_out=0 if stats_out is None else stats_out.data_ptr())


class LazyRecords:
    """Per-epoch records of an update whose numbers are still on their way from the device: the statistics buffer is copied to
    pinned host memory asynchronously and turned into python floats on first access (len() never waits).  A caller that logs every
    iteration (driver.py) sees the same values at the same place as before; a caller that does not (bench.py, users logging every
    N iterations) no longer stalls the launch queue once per iteration."""

    def __init__(self, n, host_tensors, event, build):
        self._n, self._host, self._event, self._build, self._recs = n, host_tensors, event, build, None

    def _get(self):
        if self._recs is None:
            self._event.synchronize()
            self._recs = self._build(*[h.double() for h in self._host])
            self._host = self._build = None
        return self._recs

    def __len__(self):
        return self._n

    def __iter__(self):
        return iter(self._get())

    def __getitem__(self, i):
        return self._get()[i]

    def __reduce__(self):  # pickles (torch.save) as the plain list of per-epoch dicts
        return (list, (self._get(),))


class _HostRing:
    """A few preallocated pinned staging buffers reused round-robin (a fresh pinned allocation costs ~1 ms: it must not land in
    the iteration).  A slot is reused RING iterations later; if the records that still point at it were never read they are
    materialised first (their event completed long ago)."""
    RING = 8

    def __init__(self):
        self.slots = [None] * self.RING  # (shapes, [pinned tensors], weakref to the LazyRecords using them)
        self.i = 0

    def stage(self, tensors):
        import weakref
        k = self.i
        self.i = (self.i + 1) % self.RING
        shapes = [tuple(t.shape) for t in tensors]
        slot = self.slots[k]
        if slot is not None and slot[2] is not None:
            prev = slot[2]()
            if prev is not None:
                prev._get()
        if slot is None or slot[0] != shapes:
            slot = [shapes, [torch.empty(t.shape, dtype=t.dtype, pin_memory=True) for t in tensors], None]
            self.slots[k] = slot
        for h, t in zip(slot[1], tensors):
            h.copy_(t, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        return slot[1], ev, lambda rec, slot=slot: slot.__setitem__(2, weakref.ref(rec))


def _to_host_async(ring, *tensors):
    return ring.stage(tensors)


class PPOLearner:
    """algo = "mappo" (central critic on state) or "ippo" (per-agent critic on obs)."""

    def __init__(self, algo, actor_spec, critic_spec, n_agents, hp, device, actor_params=None, critic_params=None,
                 process_group=None, world_size=1):
        assert algo in ("mappo", "ippo")
        self.lib = N.load()
        self.algo, self.A, self.hp, self.device = algo, n_agents, hp, device
        self.actor_spec, self.critic_spec = actor_spec, critic_spec
        self.pg, self.world = process_group, world_size
        if actor_params is None:
            actor_params = init_params_like_torch(actor_spec)
        if critic_params is None:
            critic_params = init_params_like_torch(critic_spec)
        self.actor = flatten_params(actor_params, device)
        self.critic = flatten_params(critic_params, device)
        Pa, Pc = actor_spec.nparams, critic_spec.nparams
        assert self.actor.numel() == Pa and self.critic.numel() == Pc, (self.actor.numel(), Pa, self.critic.numel(), Pc)
        self.opt_a = _Adam(Pa, hp.learning_rate_actor, hp.optimizer, device)
        self.opt_c = _Adam(Pc, hp.learning_rate_critic, hp.optimizer, device)
        # one flat buffer [actor grads | 8 stats | critic grads | 8 stats] = ONE all-reduce per optimiser step
        # one row per epoch (the statistics of an epoch survive the next one without copies); row 0 is `gbuf`
        self.gbuf_rows = torch.zeros(max(1, int(hp.epochs)), Pa + Pc + 2 * N.NUM_STATS, dtype=torch.float32, device=device)
        self.gbuf = self.gbuf_rows[0]
        self.g_actor = self.gbuf[:Pa + N.NUM_STATS]
        self.g_critic = self.gbuf[Pa + N.NUM_STATS:]
        self.norms = torch.zeros(2, dtype=torch.float32, device=device)
        self.ws = None    # actor pass workspace, sized on first use
        self.ws_c = None  # critic pass / value pass workspace (its own buffer: the critic epochs run on their own stream)
        # The critic's epochs only need the returns of THIS batch; the next rollout only needs the updated actor.  update() therefore
        # enqueues the critic epochs on a second stream and does not join it: the join is the first reader of the critic, i.e. the value
        # pass of the next iteration (wait_critic()).  With the rollouts double-buffered (rollout.py) the rollout of iteration i + 1
        # runs under the critic epochs of iteration i -- it is a latency chain that leaves most of the chip idle (docs/KERNEL_NOTES.md §3.5).
        self._critic_stream = None
        self._critic_done = None
        self._critic_joined = set()  # streams that already wait for _critic_done
        self.critic_span = None  # bench.py: (start, end) timing events of the last update's critic epochs (set when self.events is a list)
        # the critic's all-reduces get their own communicator: collectives issued from two streams on ONE communicator are ordered
        # by the library's internal stream, which would put the critic's message in front of the actor's next one
        # CM_FORCE_COLLECTIVES=1 (test hook): issue the all-reduces even at world size 1 -- tests/test_dist_gpu.py runs the schedules over
        # RCCL (backend "nccl") on the one GPU it has; a 1-rank sum is the identity, so the results must equal the plain run
        import os
        self._coll = world_size > 1 or (process_group is not None and os.environ.get("CM_FORCE_COLLECTIVES") == "1")
        # ... spanning exactly the ranks of `process_group` (the default group when None): new_group() must be entered by every rank of
        # the world, which env-sharded runs do (every rank builds the same learner); a learner on a sub-group gets a sub-group
        self.pg_c = process_group
        if self._coll and torch.distributed.is_initialized():
            ranks = torch.distributed.get_process_group_ranks(process_group if process_group is not None else torch.distributed.group.WORLD)
            self.pg_c = torch.distributed.new_group(ranks=ranks)
        # CM_PEER_ALLREDUCE=1 (opt-in; default: RCCL all-reduce): the [gradient | statistics] exchange of the MLP learner as a one-shot
        # peer all-reduce over hipIpc-mapped mailboxes fused with the optimiser step (dist.PeerAllReduce, csrc/cm_peer.hip)
        self.peer_a = self.peer_c = None
        if self._coll and world_size > 1 and os.environ.get("CM_PEER_ALLREDUCE") == "1" and type(self) is PPOLearner:
            self.peer_a = dist.PeerAllReduce(Pa + N.NUM_STATS, process_group)
            self.peer_c = dist.PeerAllReduce(Pc + N.NUM_STATS, process_group)
        self.global_envs = None  # env count of the WHOLE run (driver / bench set it): the schedule choice must not depend on the local shard
        self._sched_rows = {}
        self.moments = torch.zeros(3, dtype=torch.float64, device=device)
        self.values = None
        self._h0, self._h0_key = None, None  # layer-0 activations of the value pass for the first critic epoch (_keeps_h0)
        self.mom_ws = None
        self.stats_stream = None  # the stream the last update()'s statistics left on when it was NOT the launch stream (the critic's, overlapped schedules)
        self.events = None  # bench.py sets this to a list to collect per-launch (kind, start, end) HIP events
        self._ring = _HostRing()
        # optimiser steps as ONE launch (cm_optimizer_step), riding on the reduction launch of the pass when no all-reduce comes between
        # the two (cm_*_train_step); False: the stand-alone reduce / norm / update launches (A/B runs; bit-identical, tests/test_hip_parity.py)
        self.fused_step = os.environ.get("CM_FUSED_STEP", "1") != "0"

    # ------------------------------------------------------------------ helpers
    def _allreduce(self, t, pg=None):
        if self._coll:
            torch.distributed.all_reduce(t, group=self.pg if pg is None else pg)

    def _moments(self, x, ep_len, E, A, T, s):
        """(count, mean, M2) of the agent-mean over valid steps; merged across ranks (Chan et al.)."""
        if E == 0:  # an empty env shard (fewer envs than ranks): count 0 -- the merge below is still entered by every rank
            self.moments.zero_()
        else:
            need = self.lib.cm_masked_moments_workspace_bytes(E, A, T)
            if self.mom_ws is None or self.mom_ws.numel() < need:
                self.mom_ws = torch.empty(need, dtype=torch.uint8, device=self.device)
            N.check(self.lib.cm_masked_moments(N.ptr(x), N.ptr(ep_len), E, A, T, N.ptr(self.moments), N.ptr(self.mom_ws),
                                               self.mom_ws.numel(), s), "cm_masked_moments")
        dist.merge_moments_(self.moments, self.pg, self.world)

    def _empty_shard(self, b):
        """True when this rank owns no environments (an env-sharded run with fewer envs than ranks).  Such a rank launches no pass: it
        contributes all-zero [gradient | statistics] buffers (N = 0) and zero-count moments, but enters EVERY collective and applies
        every optimiser step, so its replicated parameters stay bit-identical to the other ranks'.  Without collectives an empty batch
        has no meaning: loud error."""
        if b.E > 0:
            return False
        if not self._coll or self.world <= 1:
            raise N.NativeError("empty batch (0 environments): only a rank of an env-sharded run (world size > 1) may own no environments")
        return True

    # ------------------------------------------------------------------ a6 / a7
    def compute_targets(self, b):
        """TD(lambda) returns + advantages into b.ret / b.adv (cleanmarl/mappo_multienvs.py:484-512)."""
        N.sync_env_options()
        lib, hp, s = self.lib, self.hp, N.stream_ptr()
        E, A, T = b.E, b.A, b.T
        cs = self.critic_spec
        if self._empty_shard(b):  # nothing to compute; the moment merges of the enabled normalisations are collectives: same count, same order
            self.wait_critic()
            for on, Am in ((hp.normalize_reward, 1), (hp.normalize_advantage, A), (hp.normalize_return, A)):
                if on:
                    self._moments(None, None, 0, Am, T, s)
            return
        if hp.normalize_reward:  # RolloutBuffer.get_batch, :143-146
            self._moments(b.reward, b.ep_len, E, 1, T, s)
            N.check(lib.cm_normalize(N.ptr(b.reward), N.ptr(b.ep_len), E, 1, T, N.ptr(self.moments), 1e-6, 1, s), "cm_normalize")
        Av = 1 if self.algo == "mappo" else A
        if self.values is None or self.values.shape != (E, Av, T):
            self.values = torch.empty(E, Av, T, dtype=torch.float32, device=self.device)
        x = b.state if self.algo == "mappo" else b.obs
        rows = E * T * Av
        self._ensure_ws(b)
        self.wait_critic()  # the critic epochs of the previous update ran on their own stream (under this batch's rollout)
        x_ld = b.state_ld if self.algo == "mappo" else b.obs_ld
        # "solo": this launch is ordered behind the critic stream (wait_critic above) and in front of the update -- it has the GPU to itself
        if self._keeps_h0(rows, x_ld):
            # the first critic epoch evaluates the critic on these rows with these parameters again: the value pass leaves its layer-0
            # activations behind and that epoch's one-pass kernel skips the product with W0 (critic_pass; csrc/cm_critic_fused.h, SAVED)
            if self._h0 is None or self._h0.numel() < rows * 64:
                self._h0 = torch.empty(rows * 64, dtype=torch.float32, device=self.device)
            N.check(lib.cm_value_pass_keep_h0_ld(N.ptr(x), x_ld, rows, cs.din, cs.hidden, cs.n_layers, N.ptr(self.critic), N.ptr(self.values),
                                                 N.ptr(self._h0), N.ptr(self.ws_c), self.ws_c.numel(), s), "cm_value_pass_keep_h0_ld")
            self._h0_key = (self.opt_c.step, x.data_ptr(), rows)
        else:
            self._h0_key = None
            N.check(lib.cm_mlp_forward_solo_ld(N.ptr(x), x_ld, rows, cs.din, cs.hidden, cs.n_layers, 1, N.ptr(self.critic), None,
                                               N.ptr(self.values), N.ptr(self.ws_c), self.ws_c.numel(), s), "cm_mlp_forward_solo_ld")
        N.check(lib.cm_td_lambda_scan(N.ptr(b.reward), N.ptr(self.values), N.ptr(b.ep_len), E, A, Av, T,
                                      hp.gamma, hp.td_lambda, N.ptr(b.ret), N.ptr(b.adv), s), "cm_td_lambda_scan")
        if hp.normalize_advantage:
            self._moments(b.adv, b.ep_len, E, A, T, s)
            N.check(lib.cm_normalize(N.ptr(b.adv), N.ptr(b.ep_len), E, A, T, N.ptr(self.moments), 0.0, 0, s), "cm_normalize")
        if hp.normalize_return:
            self._moments(b.ret, b.ep_len, E, A, T, s)
            N.check(lib.cm_normalize(N.ptr(b.ret), N.ptr(b.ep_len), E, A, T, N.ptr(self.moments), 0.0, 0, s), "cm_normalize")

    # ------------------------------------------------------------------ a8 - a12
    def _adam(self, params, g, opt, which, s, grad_scale=1.0, out_norm=None, stats_out=None):
        """out_norm: 1-element device view that receives the pre-clip gradient norm (default: self.norms[which]).  stats_out: see
        _Adam.next_step (the stand-alone two-launch step does not write it: its callers copy the statistics themselves)."""
        hp = self.hp
        if not self.fused_step:  # the stand-alone two-launch step (A/B runs, tests of the fused launch against it)
            opt.step += 1
            N.check(self.lib.cm_grad_norm_clip_adam(N.ptr(params), N.ptr(g), N.ptr(opt.m), N.ptr(opt.v), params.numel(), opt.step,
                                                    opt.lr, 0.9, opt.beta2, 1e-8, opt.wd, opt.kind, float(hp.clip_gradients),
                                                    grad_scale, N.ptr(self.norms[which:] if out_norm is None else out_norm), s),
                    "cm_grad_norm_clip_adam")
            return
        o = opt.next_step(params, self.norms[which:] if out_norm is None else out_norm, hp.clip_gradients, grad_scale, stats_out)
        N.check(self.lib.cm_optimizer_step(N.ptr(g), params.numel(), o, s), "cm_optimizer_step")

    def _timed(self, kind, fn, *a):
        if self.events is None:
            return fn(*a)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn(*a)
        e1.record()
        self.events.append((kind, e0, e1))

    def _keeps_h0(self, rows, x_ld):
        """Does the value pass hand its layer-0 activations to the first critic epoch?  Only where that epoch runs the one-pass kernel
        (csrc/cm_mlp_critic.hip: 65 .. 448 input columns on 16-byte rows, one hidden layer of <= 64 units, >= 131072 rows, exact fp32).  The 64
        extra floats per row pay at both ends of that range (profiles/r06_critic_h0_ab.txt: config 3's 384-wide state 7.76 -> 7.68 ms, config 4's
        115-wide observations 23.26 -> 22.78 ms although h0 is more than half an input row there).  CM_CRITIC_H0=0 / 1 forces it off / on wherever
        the kernel exists (A/B runs, tests)."""
        cs = self.critic_spec
        force = os.environ.get("CM_CRITIC_H0", "auto")
        if force == "0" or cs.kind != "mlp":
            return False
        sched = os.environ.get("CM_CRITIC_SCHEDULE", "auto")
        shape = 65 <= cs.din <= 448 and cs.hidden <= 64 and cs.n_layers == 1 and x_ld % 4 == 0 and os.environ.get("CM_MFMA", "fp32") == "fp32"
        if not shape or sched in ("split", "fused2"):
            return False
        if force == "1":
            return sched == "fused" or rows >= 131072
        return rows >= 131072

    def _ensure_ws(self, b):
        a, c = self.actor_spec, self.critic_spec
        if b.E == 0:
            return
        need = self.lib.cm_critic_workspace_bytes(b.E, b.A, b.T, 0 if self.algo == "mappo" else 1, c.din, c.hidden, c.n_layers)
        need = max(need, self.lib.cm_mlp_forward_workspace_bytes(b.E * b.T * (1 if self.algo == "mappo" else b.A), c.din, c.hidden,
                                                                 c.n_layers, 1))
        if self.ws_c is None or self.ws_c.numel() < need:
            self.wait_critic()  # the old buffer may still be in use on the critic stream
            self.ws_c = torch.empty(need, dtype=torch.uint8, device=self.device)
        if a.kind == "mlp":
            need = self.lib.cm_ppo_actor_workspace_bytes(b.E, b.A, b.T, a.din, a.hidden, a.n_layers, a.dout)
            if self.ws is None or self.ws.numel() < need:
                self.ws = torch.empty(need, dtype=torch.uint8, device=self.device)

    def critic_params(self):
        """The critic's flat parameters once the epochs still in flight on the critic stream are done (readers of `self.critic` on
        another stream -- .cpu(), clone() -- must go through here or call wait_critic() first)."""
        self.wait_critic()
        return self.critic

    def wait_critic(self):
        """Make the current stream wait for the critic epochs of the last update() (they run on their own stream and are not joined
        there).  Called by every reader of the critic parameters / optimiser state: the value pass, state_dict(), single passes."""
        if self._critic_done is not None:
            st = torch.cuda.current_stream()
            key = (st.device_index, st.cuda_stream)
            if key not in self._critic_joined:  # one wait per stream and update: a repeated wait_event is another barrier packet
                self._critic_joined.add(key)    # (kernel trace of the 512-env share: 25 us instead of 11 before the first actor pass)
                st.wait_event(self._critic_done)

    def critic_pass(self, b, s, g=None, step=None, stats=None):
        """g: [Pc + 8] gradient + statistics buffer to fill (default self.g_critic).  step: 1-element device view for the pre-clip norm --
        the critic's optimiser step then rides on the pass's reduction launch (cm_critic_train_step_ld; one process only: no all-reduce
        can come between the two)."""
        self.wait_critic()  # a no-op on the critic stream itself; any other caller must not race the epochs still in flight there
        if self._empty_shard(b):
            (self.g_critic if g is None else g).zero_()
            return
        self._ensure_ws(b)
        cs = self.critic_spec
        x = b.state if self.algo == "mappo" else b.obs
        # layer-0 activations left by the value pass: valid while neither the parameters (no optimiser step since) nor the rows have changed
        rows = b.E * b.T * (1 if self.algo == "mappo" else b.A)
        h0 = self._h0 if (self._h0_key is not None and self._h0_key == (self.opt_c.step, x.data_ptr(), rows)) else None
        if step is not None:
            o = self.opt_c.next_step(self.critic, step, self.hp.clip_gradients, stats_out=stats)
            if h0 is not None:
                N.check(self.lib.cm_critic_train_step_h0_ld(N.ptr(x), b.state_ld if self.algo == "mappo" else b.obs_ld, N.ptr(h0), N.ptr(b.ret), N.ptr(b.ep_len),
                                                            b.E, b.A, b.T, 0 if self.algo == "mappo" else 1, cs.din, cs.hidden, cs.n_layers,
                                                            N.ptr(self.g_critic if g is None else g), N.ptr(self.ws_c), self.ws_c.numel(), o, s),
                        "cm_critic_train_step_h0")
                return
            N.check(self.lib.cm_critic_train_step_ld(N.ptr(x), b.state_ld if self.algo == "mappo" else b.obs_ld, N.ptr(b.ret), N.ptr(b.ep_len),
                                                     b.E, b.A, b.T, 0 if self.algo == "mappo" else 1, cs.din, cs.hidden, cs.n_layers,
                                                     N.ptr(self.g_critic if g is None else g), N.ptr(self.ws_c), self.ws_c.numel(), o, s),
                    "cm_critic_train_step")
            return
        if h0 is not None:
            N.check(self.lib.cm_critic_fwd_bwd_h0_ld(N.ptr(x), b.state_ld if self.algo == "mappo" else b.obs_ld, N.ptr(h0), N.ptr(b.ret), N.ptr(b.ep_len),
                                                     b.E, b.A, b.T, 0 if self.algo == "mappo" else 1, cs.din, cs.hidden, cs.n_layers,
                                                     N.ptr(self.critic), N.ptr(self.g_critic if g is None else g), N.ptr(self.ws_c), self.ws_c.numel(), s),
                    "cm_critic_fwd_bwd_h0")
            return
        N.check(self.lib.cm_critic_fwd_bwd_ld(N.ptr(x), b.state_ld if self.algo == "mappo" else b.obs_ld, N.ptr(b.ret), N.ptr(b.ep_len), b.E, b.A, b.T,
                                           0 if self.algo == "mappo" else 1, cs.din, cs.hidden, cs.n_layers,
                                           N.ptr(self.critic), N.ptr(self.g_critic if g is None else g), N.ptr(self.ws_c), self.ws_c.numel(), s),
                "cm_critic_fwd_bwd")

    def actor_pass(self, b, s, g=None, step=None, stats=None):
        """g: [Pa + 8] gradient + statistics buffer to fill (default self.g_actor).  step: see critic_pass."""
        if self._empty_shard(b):
            (self.g_actor if g is None else g).zero_()
            return
        self._ensure_ws(b)
        a = self.actor_spec
        if step is not None:
            o = self.opt_a.next_step(self.actor, step, self.hp.clip_gradients, stats_out=stats)
            N.check(self.lib.cm_ppo_actor_train_step_ld(N.ptr(b.obs), b.obs_ld, N.ptr(b.avail), N.ptr(b.action), N.ptr(b.logp), N.ptr(b.adv),
                                                        N.ptr(b.ep_len), b.E, b.A, b.T, a.din, a.hidden, a.n_layers, a.dout,
                                                        self.hp.ppo_clip, self.hp.entropy_coef,
                                                        N.ptr(self.g_actor if g is None else g), N.ptr(self.ws), self.ws.numel(), o, s),
                    "cm_ppo_actor_train_step")
            return
        N.check(self.lib.cm_ppo_actor_fwd_bwd_ld(N.ptr(b.obs), b.obs_ld, N.ptr(b.avail), N.ptr(b.action), N.ptr(b.logp), N.ptr(b.adv),
                                              N.ptr(b.ep_len), b.E, b.A, b.T, a.din, a.hidden, a.n_layers, a.dout,
                                              N.ptr(self.actor), self.hp.ppo_clip, self.hp.entropy_coef,
                                              N.ptr(self.g_actor if g is None else g), N.ptr(self.ws), self.ws.numel(), s),
                "cm_ppo_actor_fwd_bwd")

    def overlap_critic(self, b):
        """Schedule of update(): 0 = both networks interleaved on the current stream; 2 = the critic's epochs on a second,
        LOWEST-priority stream released at the start of the update and not joined there -- its kernels take the compute units
        the actor's kernels and, above all, the NEXT rollout leave idle (the rollout is a latency chain on a fraction of the chip);
        1 = the same stream released only when the actor's epochs are done (wins for the smallest batches, where every kernel
        leaves most of the chip idle and the critic's launches only delay the actor's).
        Measured (profiles/r02_critic_overlap_schedules.txt, ms per iteration, schedule 0 / 1 / 2): config 3 at 512 envs (one GPU's
        share of 8) 1.81 / 1.60 / 1.56, 1024 envs 2.92 / 2.98 / 2.77, 4096 envs 9.51 / 9.53 / 9.31; config 4 at 256 envs 4.10 / 4.19 /
        3.91; config 2 1.44 / 1.30 / 1.32; config 3 at 256 envs (a share of 16) 1.26 / 1.09 / 1.27.  Re-measured at the end of round 3, once
        the optimiser-step launch and the streaming dW0 kernel could be placed beside the other stream's kernels (gpurun_out/r03ac, r03ad;
        schedule 0 / 1 / 2): config 3 at 2048 envs 4.72 / 4.74 / 4.80, 1024 envs 2.71 / 2.74 / 2.68, 768 envs 2.35 / 2.19 / 2.09, 384 envs
        - / 1.26 / 1.24, 256 envs - / 0.97 / 1.00, 128 envs - / 0.745 / 0.752; config 2 1.33 / 1.21 / 1.19 (512 envs: - / 0.885 / 0.870);
        config 4 at 512 envs 6.22 / 6.26 / 5.93, 256 envs 3.26 / 3.29 / 3.07, 128 envs - / 1.79 / 1.60.  Default: 1 below 2^17 rows, 2 from
        there on.  At full size (>= 2^21 rows) schedule 2 is the faster one as well -- round 6, on the round-5 kernels: 7.905 vs 7.985 ms per
        iteration at 4096 envs (profiles/r06_schedule_full_size_ab.txt; the critic's one-pass kernel needs a whole CU's registers, so beside the
        actor's persistent workgroups it only fills the tails of the actor's launches: 1 %, not the 2.6 % of round 2's kernels) -- and ships;
        rounds 2 - 5 kept schedule 0 there so that the actor kernel, the one the roofline is quoted on, was timed alone: bench.py now times that
        kernel in a SOLO leg after the timed region (roofline.source) and reports the in-iteration duration beside it.
        CM_CRITIC_OVERLAP=0 / 1 / 2 forces a schedule."""
        import os
        v = os.environ.get("CM_CRITIC_OVERLAP")
        if v in ("0", "1", "2"):
            return int(v)
        rows = self._schedule_rows(b)
        return 1 if rows < (1 << 17) else 2

    def _schedule_rows(self, b):
        """Row count the schedule is chosen from -- the SAME number on every rank: env shards may differ by one env (dist.shard), and
        ranks on different sides of a threshold would interleave their collectives differently (schedule 0 waits for the critic's
        message before the next actor pass, schedule 1 issues it after the actor's epochs: a cross-rank deadlock).  With
        `global_envs` set (driver, bench) it is the largest shard's; otherwise the maximum over the ranks, agreed once per batch shape."""
        if not self._coll or self.world <= 1:
            return b.E * b.A * b.T
        if self.global_envs is not None:
            return -(-int(self.global_envs) // self.world) * b.A * b.T
        key = (b.E, b.A, b.T)
        if key not in self._sched_rows:
            t = torch.tensor([b.E * b.A * b.T], dtype=torch.int64, device=self.device)
            torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX, group=self.pg)
            self._sched_rows[key] = int(t.item())
        return self._sched_rows[key]

    def update(self, b, keep_grads=False):
        """`epochs` full-batch PPO steps (cleanmarl/mappo_multienvs.py:521-594).  The two networks are independent (separate losses,
        separate optimisers, :341-343): per network the sums and the order of its optimiser steps are those of the reference's
        interleaved loop, whichever schedule runs:
          * large batch: actor pass, critic pass, Adam, Adam per epoch on the current stream; N > 1: each network's all-reduce is
            issued right after its pass and waited for only where that network's optimiser step is due (the actor's message travels
            under the critic pass, the critic's under the next epoch's actor pass);
          * small batch (overlap_critic): the actor's epochs on the current stream, then the critic's epochs on a second stream that
            is NOT joined here -- the caller's next rollout overlaps them, the next value pass waits for them (wait_critic()).
        Returns per-epoch records (LazyRecords: no host wait)."""
        N.sync_env_options()
        hp, s = self.hp, N.stream_ptr()
        Pa, Pc = self.actor.numel(), self.critic.numel()
        nE0 = int(hp.epochs)
        if self.gbuf_rows.shape[0] < nE0:
            raise N.NativeError(f"epochs={nE0} exceeds the {self.gbuf_rows.shape[0]} gradient rows allocated at construction")
        self._ensure_ws(b)
        self.wait_critic()
        main = torch.cuda.current_stream()
        overlap = 0 if keep_grads else self.overlap_critic(b)
        # one [actor grads | 8 | critic grads | 8] row per epoch: the statistics survive the next epoch without per-epoch copies,
        # the norms are written by the Adam kernel straight into `rec`; row 0 is self.gbuf (what single passes and tests read)
        # [actor stats | critic stats | actor norm | critic norm] per epoch.  Two persistent buffers used alternately (every field is
        # written by each update: no fill launch; the previous update's asynchronous copy to the host may still be reading the other one)
        self._rec_i = getattr(self, "_rec_i", 0) ^ 1
        recs_ = getattr(self, "_recs", None)
        if recs_ is None or recs_[0].shape[0] != nE0:
            recs_ = self._recs = [torch.zeros(nE0, 2 * N.NUM_STATS + 2, dtype=torch.float32, device=self.device) for _ in range(2)]
        rec = recs_[self._rec_i]
        kept_a, kept_c = [], []
        timed = self.events is not None
        self.critic_span = None

        ride = self.fused_step and not self._coll  # the optimiser step rides on the pass's reduction launch
        # statistics of an epoch: written into `rec` by the step launch itself (stats_out) -- except by the stand-alone A/B step and by
        # the passes of a rank without environments (no launch at all: their zero buffers are copied like before)
        copy_stats = (not self.fused_step) or b.E == 0

        def actor_step(ep, wa):
            g_actor = self.gbuf_rows[ep][:Pa + N.NUM_STATS]
            if ride:
                if keep_grads:
                    kept_a.append((g_actor[:Pa].clone(), self.actor.clone()))
                return
            if self.peer_a is not None:
                self.peer_a.step(g_actor, Pa, self.opt_a.next_step(self.actor, rec[ep, 2 * N.NUM_STATS:], hp.clip_gradients, stats_out=rec[ep, :N.NUM_STATS]), s)
            else:
                if wa is not None:
                    wa.wait()
                self._adam(self.actor, g_actor, self.opt_a, 0, s, out_norm=rec[ep, 2 * N.NUM_STATS:], stats_out=rec[ep, :N.NUM_STATS])
            if keep_grads:
                kept_a.append((g_actor[:Pa].clone(), self.actor.clone()))

        def critic_step(ep, wc, sc):
            g_critic = self.gbuf_rows[ep][Pa + N.NUM_STATS:]
            if ride:
                if keep_grads:
                    kept_c.append((g_critic[:Pc].clone(), self.critic.clone()))
                return
            if self.peer_c is not None:
                self.peer_c.step(g_critic, Pc, self.opt_c.next_step(self.critic, rec[ep, 2 * N.NUM_STATS + 1:], hp.clip_gradients,
                                                                     stats_out=rec[ep, N.NUM_STATS:2 * N.NUM_STATS]), sc)
            else:
                if wc is not None:
                    wc.wait()
                self._adam(self.critic, g_critic, self.opt_c, 1, sc, out_norm=rec[ep, 2 * N.NUM_STATS + 1:], stats_out=rec[ep, N.NUM_STATS:2 * N.NUM_STATS])
            if keep_grads:
                kept_c.append((g_critic[:Pc].clone(), self.critic.clone()))

        self.stats_stream = None
        if not overlap:
            pending = None  # the critic's optimiser step of the previous epoch, due before the next critic pass
            for ep in range(nE0):
                g = self.gbuf_rows[ep]
                self._timed("actor", self.actor_pass, b, s, g[:Pa + N.NUM_STATS], rec[ep, 2 * N.NUM_STATS:] if ride else None, rec[ep, :N.NUM_STATS])
                wa = dist.allreduce_sum_async(g[:Pa + N.NUM_STATS], self.pg) if (self._coll and self.peer_a is None) else None
                if pending is not None:
                    critic_step(*pending, s)
                self._timed("critic", self.critic_pass, b, s, g[Pa + N.NUM_STATS:], rec[ep, 2 * N.NUM_STATS + 1:] if ride else None,
                            rec[ep, N.NUM_STATS:2 * N.NUM_STATS])
                wc = dist.allreduce_sum_async(g[Pa + N.NUM_STATS:], self.pg_c) if (self._coll and self.peer_c is None) else None
                actor_step(ep, wa)
                if self._coll:
                    pending = (ep, wc)
                else:
                    critic_step(ep, None, s)
            if pending is not None:
                critic_step(*pending, s)
            if copy_stats:  # (the fused step launches wrote every epoch's statistics into `rec` themselves: cm_opt_step_t::stats_out)
                rec[:, :N.NUM_STATS] = self.gbuf_rows[:nE0, Pa:Pa + N.NUM_STATS]
                rec[:, N.NUM_STATS:2 * N.NUM_STATS] = self.gbuf_rows[:nE0, Pa + N.NUM_STATS + Pc:]
            host, ev, attach = _to_host_async(self._ring, rec)  # no host wait here: see LazyRecords
        else:
            if self._critic_stream is None:
                self._critic_stream = N.low_priority_stream(self.device)
            side = self._critic_stream

            def actor_epoch(ep):
                g_actor = self.gbuf_rows[ep][:Pa + N.NUM_STATS]
                self._timed("actor", self.actor_pass, b, s, g_actor, rec[ep, 2 * N.NUM_STATS:] if ride else None, rec[ep, :N.NUM_STATS])
                actor_step(ep, dist.allreduce_sum_async(g_actor, self.pg) if (self._coll and self.peer_a is None) else None)

            def critic_epoch(ep):
                with torch.cuda.stream(side):
                    sc = N.stream_ptr()
                    if timed and ep == 0:
                        self._c0 = torch.cuda.Event(enable_timing=True)
                        self._c0.record()
                    g_critic = self.gbuf_rows[ep][Pa + N.NUM_STATS:]
                    self._timed("critic", self.critic_pass, b, sc, g_critic, rec[ep, 2 * N.NUM_STATS + 1:] if ride else None,
                                rec[ep, N.NUM_STATS:2 * N.NUM_STATS])
                    critic_step(ep, dist.allreduce_sum_async(g_critic, self.pg_c) if (self._coll and self.peer_c is None) else None, sc)
                    if ep == nE0 - 1:
                        if copy_stats:
                            rec[:, N.NUM_STATS:2 * N.NUM_STATS] = self.gbuf_rows[:nE0, Pa + N.NUM_STATS + Pc:]
                        if timed:
                            c1 = torch.cuda.Event(enable_timing=True)
                            c1.record()
                            self.critic_span = (self._c0, c1)

            if overlap == 2:
                # released now, lowest priority: the critic's kernels take the compute units the actor's kernels and the next rollout leave
                # idle.  HOST order matters as much as stream order: the launches are interleaved epoch by epoch, actor first -- with all of
                # the critic's ~20 launches enqueued ahead of the first actor kernel the main stream sat idle for ~150 us per iteration
                # (kernel trace of the 512-env share) while the critic ran alone
                side.wait_stream(main)
                for ep in range(nE0):
                    actor_epoch(ep)
                    critic_epoch(ep)
            else:  # released when the actor's epochs are done: they run under the next rollout only
                for ep in range(nE0):
                    actor_epoch(ep)
                side.wait_stream(main)
                for ep in range(nE0):
                    critic_epoch(ep)
            side.wait_stream(main)  # the statistics need both halves; they leave on the critic stream, `main` never waits for them
            with torch.cuda.stream(side):
                if copy_stats:
                    rec[:, :N.NUM_STATS] = self.gbuf_rows[:nE0, Pa:Pa + N.NUM_STATS]  # on the side stream too: nothing trails the actor's last step on `main`
                host, ev, attach = _to_host_async(self._ring, rec)
                self._critic_done = torch.cuda.Event()
                self._critic_done.record(side)
                self.stats_stream = side
                self._critic_joined = {(side.device_index, side.cuda_stream)}  # the critic stream itself is ordered behind its own work
        nE, ent_coef = int(hp.epochs), hp.entropy_coef
        kept = [(ka[0], kc[0], ka[1], kc[1]) for ka, kc in zip(kept_a, kept_c)]

        peers = [p for p in (self.peer_a, self.peer_c) if p is not None]

        def build(r):
            for p in peers:  # the statistics have arrived, so every step of this update has run: did one give up waiting for its peers?
                p.check()
            out = []
            for ep in range(nE):
                st_a, st_c = r[ep, :N.NUM_STATS], r[ep, N.NUM_STATS:2 * N.NUM_STATS]
                n = float(st_a[N.STAT_COUNT])
                d = dict(actor_loss=float(-st_a[N.STAT_PG] - ent_coef * st_a[N.STAT_ENT]) / n,
                         critic_loss=float(st_c[N.STAT_VLOSS]) / float(st_c[N.STAT_COUNT]),
                         entropy=float(st_a[N.STAT_ENT]) / n, kl=float(st_a[N.STAT_KL]) / n,
                         clipfrac=float(st_a[N.STAT_CLIP]) / n,
                         actor_gnorm=float(r[ep, 2 * N.NUM_STATS]), critic_gnorm=float(r[ep, 2 * N.NUM_STATS + 1]), n_valid=n)
                if keep_grads:
                    d.update(actor_grads=kept[ep][0], critic_grads=kept[ep][1], actor_after=kept[ep][2], critic_after=kept[ep][3])
                out.append(d)
            return out
        out = LazyRecords(nE, host, ev, build)
        attach(out)
        return out

    def train_iteration(self, b, keep_grads=False):
        self.compute_targets(b)
        return self.update(b, keep_grads=keep_grads)

    def close(self):
        """Release what outlives a learner otherwise: the hipIpc mappings / mailboxes of the opt-in peer all-reduce (dist.PeerAllReduce)."""
        for name in ("peer_a", "peer_c"):
            peer = getattr(self, name, None)
            if peer is not None:
                torch.cuda.synchronize(self.device)  # no step launch may still be reading a mailbox
                peer.close()
                setattr(self, name, None)

    # ------------------------------------------------------------------ checkpointing (README TODO of the reference)
    def state_dict(self):
        """Flat parameters + Adam moments + step counters (CPU tensors; torch.save-able)."""
        self.wait_critic()
        hp = {k: getattr(self.hp, k) for k in ("optimizer", "learning_rate_actor", "learning_rate_critic", "epochs", "tbptt")}
        return dict(algo=self.algo, actor_spec=vars(self.actor_spec), critic_spec=vars(self.critic_spec), hp=hp,
                    actor=self.actor.cpu(), critic=self.critic.cpu(),
                    opt_a=dict(m=self.opt_a.m.cpu(), v=self.opt_a.v.cpu(), step=self.opt_a.step),
                    opt_c=dict(m=self.opt_c.m.cpu(), v=self.opt_c.v.cpu(), step=self.opt_c.step))

    def load_state_dict(self, sd):
        if sd["algo"] != self.algo or sd["actor_spec"] != vars(self.actor_spec) or sd["critic_spec"] != vars(self.critic_spec):
            raise N.NativeError("checkpoint was written for a different algorithm / network shape")
        saved = sd.get("hp")
        if saved is not None and saved.get("optimizer") != self.hp.optimizer:  # the moments of one optimiser mean nothing to another
            raise N.NativeError(f"checkpoint holds {saved.get('optimizer')} state, this run uses --optimizer={self.hp.optimizer}")
        if saved is not None:
            diff = {k: (v, getattr(self.hp, k)) for k, v in saved.items() if k != "optimizer" and getattr(self.hp, k) != v}
            if diff:
                import warnings
                warnings.warn(f"resuming with different hyper-parameters than the checkpoint was written with: {diff}")
        self.wait_critic()
        self.actor.copy_(sd["actor"]); self.critic.copy_(sd["critic"])
        self._h0_key = None
        for opt, o in ((self.opt_a, sd["opt_a"]), (self.opt_c, sd["opt_c"])):
            opt.m.copy_(o["m"]); opt.v.copy_(o["v"]); opt.step = int(o["step"])
