"""GRU-actor variants (reference: cleanmarl/mappo_lstm_multienvs.py, cleanmarl/ippo_lstm_multienvs.py).

GRUPPOLearner keeps the critic path of PPOLearner and replaces the actor update by the truncated-BPTT
schedule of cleanmarl/mappo_lstm_multienvs.py:551-664: per epoch, h = None; for every chunk of `tbptt` steps
one fused forward + backward-through-time launch (cm_gru_actor_chunk_fwd_bwd), one all-reduce (N > 1) and one
Adam step with the chunk loss normalised by N_chunk * T_chunk (:605-607); h is carried detached (:620); the
critic takes one step per epoch (:646-655).
"""
import ctypes as C
import os

import torch

from . import _native as N
from . import dist
from .learner import LazyRecords, _to_host_async, DeviceBatch, PPOLearner

_GOLD = 0x9E3779B97F4A7C15


class GRUPPOLearner(PPOLearner):
    def __init__(self, algo, actor_spec, critic_spec, n_agents, hp, device, actor_params=None, critic_params=None,
                 process_group=None, world_size=1):
        assert actor_spec.kind == "gru"
        super().__init__(algo, actor_spec, critic_spec, n_agents, hp, device, actor_params, critic_params, process_group, world_size)
        self.gru_ws = None
        self.g_rows = None
        self.h = [None, None]
        self._critic_stream = None
        # where the critic's epochs run (the critic stream's own order never changes, so the results do not depend on it):
        #   "beside"  epoch e on a second normal-priority stream beside actor epoch e, joined at the end of update() (rounds 1 - 3);
        #   "low"     the same on the lowest-priority stream;
        #   "deferN"  the last N epochs behind the actor's epochs on the lowest-priority stream: they run under the NEXT rollout and are
        #             joined by the next value pass (wait_critic), the first epochs stay beside the actor's.
        # Kernel trace of config 5 (rocprofv3 --kernel-trace; today: tools/gpu/run.sh "timeline --workload cfg5"): a critic epoch (pass 143 + streamed dW0 142 + step 35 us) beside the
        # pipelined GRU sweeps -- which have no idle CUs any more -- stretches the first chunk of the actor epoch by ~115 us (forward
        # sweep 67 -> 105 us, backward 80 -> 157); beside the rollout it costs 69 us once, however many epochs follow.  Measured
        # (CM_GRU_CRITIC sweep; today: tools/gpu/run.sh "ab CM_GRU_CRITIC beside defer1 defer2 defer3 -- --workload cfg5", two runs each): beside 7.33 / 7.34 ms, defer1 7.20 / 7.23, defer2 7.23 / 7.26 (the value pass starts
        # to wait), defer3 7.30.  Round 4, after the state rows were padded to 16 bytes (one-pass critic, 160 -> 111 us alone: two epochs now
        # fit under the 0.78 ms rollout without the value pass waiting; the same sweep in round 4, three runs each): defer1 7.15 / 7.18 / 7.15,
        # defer2 7.12 / 7.11 / 7.10, defer3 7.16 (the value pass waits 0.1 ms).  CM_GRU_CRITIC is a Python-side A/B hook, not an option of
        # the C-ABI.
        self.critic_schedule = os.environ.get("CM_GRU_CRITIC", "defer2")
        self._parse_schedule(self.critic_schedule)  # a typo fails here, not in the middle of an update
        self._sched_fixed = None
        self._grecs, self._grec_i = None, 0

    @staticmethod
    def _parse_schedule(sched):
        """-> (stream kind, number of deferred epochs or None for "all"); raises on anything but beside | low | defer | defer<N>."""
        if sched in ("beside", "low"):
            return sched, 0
        if sched.startswith("defer") and (sched[5:] == "" or sched[5:].isdigit()):
            return "low", (int(sched[5:]) if sched[5:] else None)
        raise N.NativeError(f"critic_schedule / CM_GRU_CRITIC = {sched!r}: expected beside, low, defer or defer<N>")

    def _ensure(self, b):
        a = self.actor_spec
        need = self.lib.cm_gru_workspace_bytes(b.E, b.A, a.din, a.hidden, a.dout, int(self.hp.tbptt))
        if self.gru_ws is None or self.gru_ws.numel() < need:
            self.gru_ws = torch.empty(need, dtype=torch.uint8, device=self.device)
        if self.h[0] is None or self.h[0].shape[0] != b.E * b.A:
            self.h = [torch.zeros(b.E * b.A, a.hidden, dtype=torch.float32, device=self.device) for _ in range(2)]

    def update(self, b, keep_grads=False):
        N.sync_env_options()
        hp, s, a = self.hp, N.stream_ptr(), self.actor_spec
        empty = self._empty_shard(b)  # a rank without environments: zero buffers, every collective and step (learner.PPOLearner._empty_shard)
        if not empty:
            self._ensure(b)
        Pa, Pc = self.actor.numel(), self.critic.numel()
        T, tb = b.T, int(hp.tbptt)
        chunks = [(t0, min(t0 + tb, T)) for t0 in range(0, T, tb)]
        nE = int(hp.epochs)
        # statistics of this update: two persistent buffer pairs used alternately (like PPOLearner._recs).  With deferred critic epochs
        # the un-joined critic stream still writes rec_c and copies both to the host after update() has returned: buffers allocated per
        # update on the launch stream would go back to its allocator pool with that work pending.  Every field is rewritten by
        # each update (all epochs, all chunks), and the pair used two updates ago has been read: its copy precedes the previous update's
        # critic epochs on the critic stream, which this update's wait_critic() below joins before anything is enqueued.
        if self._grecs is None or self._grecs[0][0].shape[:2] != (nE, len(chunks)):
            self._grecs = [(torch.zeros(nE, len(chunks), N.NUM_STATS + 1, dtype=torch.float32, device=self.device),
                            torch.zeros(nE, N.NUM_STATS + 1, dtype=torch.float32, device=self.device)) for _ in range(2)]
        self._grec_i ^= 1
        rec_a, rec_c = self._grecs[self._grec_i]
        if self.g_rows is None or self.g_rows.shape[0] < len(chunks):
            self.g_rows = torch.zeros(len(chunks), Pa + N.NUM_STATS, dtype=torch.float32, device=self.device)
        kept, kept_c = [], []
        # The critic's epoch (one pass + one step, independent of the actor: it reads returns and states only, own workspace, own
        # gradient buffer) is enqueued on a second stream: at the start of an actor epoch, or -- the last `n_defer` of them -- behind the
        # actor's epochs, i.e. under the next rollout (self.critic_schedule, measurements in __init__).  The serial chain of TBPTT chunk
        # kernels is never extended by them.  Issue order is identical on all ranks, so the collectives still pair up.
        ride = self.fused_step and not self._coll
        # the fused step launch writes an epoch's / a chunk's statistic sums and norm straight into the record buffers (cm_opt_step_t::
        # stats_out, out_norm): no copy launch per epoch; the stand-alone A/B step and a rank without environments copy as before
        direct = self.fused_step and not empty
        main = torch.cuda.current_stream()
        kind, defer = self._parse_schedule(self.critic_schedule)
        if self._critic_stream is None:
            # one per process: see _native.low_priority_stream.  The stream kind is fixed by the first update
            self._critic_stream = N.side_stream(self.device) if kind == "beside" else N.low_priority_stream(self.device)
            self._sched_fixed = kind
        elif kind != self._sched_fixed:
            raise N.NativeError(f"critic_schedule changed from a {self._sched_fixed!r} to a {kind!r} stream after the first update")
        side = self._critic_stream
        self.wait_critic()
        self._critic_done = None
        side.wait_stream(main)

        def critic_epoch(ep):
            with torch.cuda.stream(side):
                sc = N.stream_ptr()
                if ride:  # one process: the optimiser step rides on the pass's reduction launch
                    self._timed("critic", self.critic_pass, b, sc, None, rec_c[ep, N.NUM_STATS:], rec_c[ep, :N.NUM_STATS])
                else:
                    self._timed("critic", self.critic_pass, b, sc)
                    self._allreduce(self.g_critic, self.pg_c)  # own communicator: never queues ahead of a chunk's message
                    self._adam(self.critic, self.g_critic, self.opt_c, 1, sc, out_norm=rec_c[ep, N.NUM_STATS:],
                               stats_out=rec_c[ep, :N.NUM_STATS] if direct else None)
                if not direct:
                    rec_c[ep, :N.NUM_STATS] = self.g_critic[Pc:]
                if keep_grads:
                    kept_c.append((self.g_critic[:Pc].clone(), self.critic.clone()))

        n_defer = min(nE, nE if defer is None else defer)
        for ep in range(nE):
            if ep < nE - n_defer:
                critic_epoch(ep)
            steps = []
            h_in = None
            if self.events is not None:  # bench.py: one event pair around all TBPTT chunks of the epoch
                ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                ev0.record()
            for ci, (t0, t1) in enumerate(chunks):
                h_out = self.h[ci & 1]
                g = self.g_rows[ci]  # one [grads | stats] row per chunk: the statistics survive without per-chunk copies
                if ride:
                    o = self.opt_a.next_step(self.actor, rec_a[ep, ci, N.NUM_STATS:], hp.clip_gradients, 1.0 / (t1 - t0),
                                             stats_out=rec_a[ep, ci, :N.NUM_STATS])
                    N.check(self.lib.cm_gru_actor_chunk_train_step(
                        N.ptr(b.obs), N.ptr(b.avail), N.ptr(b.action), N.ptr(b.logp), N.ptr(b.adv), N.ptr(b.ep_len),
                        b.E, b.A, T, t0, t1, a.din, a.hidden, a.dout, N.ptr(h_in), N.ptr(h_out),
                        hp.ppo_clip, hp.entropy_coef, N.ptr(g), N.ptr(self.gru_ws), self.gru_ws.numel(), o, s),
                        "cm_gru_actor_chunk_train_step")
                elif empty:
                    g.zero_()
                    self._allreduce(g)
                    self._adam(self.actor, g, self.opt_a, 0, s, grad_scale=1.0 / (t1 - t0), out_norm=rec_a[ep, ci, N.NUM_STATS:])
                else:
                    N.check(self.lib.cm_gru_actor_chunk_fwd_bwd(
                        N.ptr(b.obs), N.ptr(b.avail), N.ptr(b.action), N.ptr(b.logp), N.ptr(b.adv), N.ptr(b.ep_len),
                        b.E, b.A, T, t0, t1, a.din, a.hidden, a.dout, N.ptr(self.actor), N.ptr(h_in), N.ptr(h_out),
                        hp.ppo_clip, hp.entropy_coef, N.ptr(g), N.ptr(self.gru_ws), self.gru_ws.numel(), s),
                        "cm_gru_actor_chunk_fwd_bwd")
                    self._allreduce(g)
                    self._adam(self.actor, g, self.opt_a, 0, s, grad_scale=1.0 / (t1 - t0), out_norm=rec_a[ep, ci, N.NUM_STATS:],
                               stats_out=rec_a[ep, ci, :N.NUM_STATS] if direct else None)
                if keep_grads:
                    steps.append((g[:Pa].clone(), self.actor.clone()))
                h_in = h_out
            if not direct:
                rec_a[ep, :, :N.NUM_STATS] = self.g_rows[:len(chunks), Pa:]  # one strided copy per epoch
            if self.events is not None:
                ev1.record()
                self.events.append(("actor", ev0, ev1))
            if keep_grads:
                kept.append((steps,))
        if n_defer:
            # all critic epochs behind the actor's, on the lowest-priority stream, NOT joined here: they run under the next rollout (one
            # six-wave workgroup on 171 of the 256 CUs at config 5) and are joined by the next value pass (wait_critic)
            side.wait_stream(main)
            for ep in range(nE - n_defer, nE):
                critic_epoch(ep)
            with torch.cuda.stream(side):
                host, ev, attach = _to_host_async(self._ring, rec_a, rec_c)
                self._critic_done = torch.cuda.Event()
                self._critic_done.record(side)
                self._critic_joined = {(side.device_index, side.cuda_stream)}
            self.stats_stream = side
        else:
            main.wait_stream(side)
            self.stats_stream = None
        if keep_grads:
            kept = [k + kept_c[ep] for ep, k in enumerate(kept)]
        ent_coef = hp.entropy_coef

        def build(ra, rc):
            out = []
            for ep in range(nE):
                tot = ra[ep, :, :N.NUM_STATS].sum(0)
                n = float(tot[N.STAT_COUNT])
                d = dict(actor_loss=float(-tot[N.STAT_PG] - ent_coef * tot[N.STAT_ENT]) / n,
                         critic_loss=float(rc[ep, N.STAT_VLOSS]) / float(rc[ep, N.STAT_COUNT]),
                         entropy=float(tot[N.STAT_ENT]) / n, kl=float(tot[N.STAT_KL]) / n, clipfrac=float(tot[N.STAT_CLIP]) / n,
                         actor_gnorm=float(ra[ep, :, N.NUM_STATS].mean()), critic_gnorm=float(rc[ep, N.NUM_STATS]), n_valid=n)
                if keep_grads:
                    d.update(actor_steps=kept[ep][0], critic_grads=kept[ep][1], critic_after=kept[ep][2])
                out.append(d)
            return out
        if not n_defer:
            host, ev, attach = _to_host_async(self._ring, rec_a, rec_c)  # no host wait here: see learner.LazyRecords
        out = LazyRecords(nE, host, ev, build)
        attach(out)
        return out


class GRUSyntheticRollout:
    """Synthetic MPE-like env with the GRU actor: T x (cm_gru_policy_act, cm_synth_env_step), hidden state on
    device (reference rollout: cleanmarl/mappo_lstm_multienvs.py:392-479, h = None at the start of each episode)."""

    def __init__(self, E, A, T, seed=1, agent_ids=True, device="cuda:0", env_offset=0, pad_state=True):
        """pad_state=False: contiguous state rows from the start -- for rollouts that will only ever take the per-step path (greedy
        evaluation, evaluate.DeviceEvaluator), which would otherwise re-allocate its buffers behind a device synchronisation on first use."""
        self.lib = N.load()
        self.E, self.A, self.T, self.K = E, A, T, 5
        self.agent_ids = bool(agent_ids)
        self.Do, self.Ds = 6 * A + (A if agent_ids else 0), 6 * A * A
        self.seed, self.env_offset, self.device = int(seed), int(env_offset), torch.device(device)
        # two buffers used alternately: the learner's critic epochs (own stream, learner.py) may still read episode i's states and
        # returns while episode i + 1 is being written
        # the STATE rows are padded to a multiple of 4 floats (the critic's passes then read 16-byte aligned rows: 150 -> 152 floats at 5 agents,
        # critic epoch 160 -> 111 us at config 5); the observations stay contiguous (the recurrent kernels read them as such)
        self.batches = [DeviceBatch(E, A, T, self.Do, self.Ds, self.K, self.device, pad_state=bool(pad_state)) for _ in range(2)]
        for bb in self.batches:
            bb.avail.fill_(1)
            bb.ep_len.fill_(T)
        self.batch = self.batches[0]  # the most recently collected one
        self.env_state = torch.zeros(E, 6 * A, dtype=torch.float32, device=self.device)
        self.h = None
        self.episode = 0

    def collect(self, actor_flat, actor_spec, fused=None, eps=0.0):
        """eps < 0: greedy actions (evaluation rollouts of --greedy_eval, evaluate.py) -- on the per-step path, whose act kernel has the
        argmax; the recurrent scripts have no epsilon-mixed exploration (eps > 0 is an error of the library)."""
        N.sync_env_options()
        self.batch = self.batches[self.episode & 1]
        lib, b, s = self.lib, self.batch, N.stream_ptr()
        E, A, T, Do, K = self.E, self.A, self.T, self.Do, self.K
        can_fuse = bool(lib.cm_gru_rollout_spread_supported(A, int(self.agent_ids), actor_spec.hidden))
        if eps != 0.0:
            fused = False
        if fused is None:
            fused = can_fuse
        if fused:  # the whole episode in one persistent launch (same seeds => same rollout as the per-step path below)
            if not can_fuse:
                raise N.NativeError("fused GRU rollout requested for an unsupported shape")
            act_seed = (self.seed + (self.episode + 1) * _GOLD) & 0xFFFFFFFFFFFFFFFF
            N.check(lib.cm_gru_rollout_spread_ld(N.ptr(self.env_state), E, A, T, int(self.agent_ids), self.seed, act_seed, self.env_offset,
                                                 self.episode, N.ptr(actor_flat), actor_spec.hidden, N.ptr(b.obs), N.ptr(b.state), b.state_ld,
                                                 N.ptr(b.action), N.ptr(b.logp), N.ptr(b.reward), s), "cm_gru_rollout_spread_ld")
            self.episode += 1
            return b
        if b.state_ld != b.Ds:
            # the per-step env kernels (cm_synth_env_reset / _step) write contiguous rows: shapes the fused kernel does not cover, greedy
            # evaluation rollouts and explicit fused=False runs switch this rollout to unpadded buffers, once (rollout.py does the same).
            # The learner's critic epochs may still read the old buffers on their own stream: wait before their storage is released
            torch.cuda.synchronize(self.device)
            self.batches = [DeviceBatch(E, A, T, self.Do, self.Ds, self.K, self.device) for _ in range(2)]
            for bb in self.batches:
                bb.avail.fill_(1)
                bb.ep_len.fill_(T)
            self.batch = b = self.batches[self.episode & 1]
        if self.h is None:
            self.h = torch.zeros(E * A, actor_spec.hidden, dtype=torch.float32, device=self.device)
        self.h.zero_()
        N.check(lib.cm_synth_env_reset(N.ptr(self.env_state), E, A, int(self.agent_ids), self.seed, self.env_offset,
                                       self.episode, N.ptr(b.obs), N.ptr(b.state), T, s), "cm_synth_env_reset")
        act_seed = (self.seed + (self.episode + 1) * _GOLD) & 0xFFFFFFFFFFFFFFFF
        off = lambda t, nbytes: C.c_void_p(t.data_ptr() + nbytes)
        need = lib.cm_gru_policy_act_workspace_bytes(E * A, actor_spec.din, actor_spec.hidden, K)  # used by the layered shapes only
        if getattr(self, "act_ws", None) is None or self.act_ws.numel() < need:
            self.act_ws = torch.empty(need, dtype=torch.uint8, device=self.device)
        for t in range(T):
            N.check(lib.cm_gru_policy_act_ws(off(b.obs, 4 * t * Do), T * Do, off(b.avail, t * K), T * K, E * A, actor_spec.din,
                                             actor_spec.hidden, K, N.ptr(actor_flat), N.ptr(self.h), float(eps), act_seed, self.env_offset * A, t,
                                             off(b.action, 4 * t), off(b.logp, 4 * t), T, N.ptr(self.act_ws), self.act_ws.numel(), s),
                    "cm_gru_policy_act_ws")
            N.check(lib.cm_synth_env_step(N.ptr(self.env_state), N.ptr(b.action), E, A, int(self.agent_ids), t, T,
                                          N.ptr(b.reward), N.ptr(b.obs), N.ptr(b.state), s), "cm_synth_env_step")
        self.episode += 1
        return b
