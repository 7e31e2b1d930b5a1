"""Evaluation block of the scripts (cleanmarl/mappo_multienvs.py:614-650; GRU: mappo_lstm_multienvs.py:675-714; COMA:
coma_multienvs.py:692-727) as BATCHED episodes.

The reference plays ``num_eval_ep`` episodes one after the other on one extra env, one ``actor.act`` per step.  Episodes are
independent, so here they run side by side:

* ``DeviceEvaluator`` (``--env_type=synthetic*``): the ``num_eval_ep`` episodes are ONE device rollout (cm_rollout_spread_ld /
  cm_gru_rollout_spread / the shape env's one-launch act pass; ``--greedy_eval``: eps < 0) over the env indices
  ``EVAL_ENV_BASE + j`` on a snapshot of the actor's parameters, enqueued on its own lowest-priority stream: it runs under the next
  iteration's kernels, nothing on the training streams waits for it, and the three logged numbers leave through page-locked memory
  and are read when the iteration is accounted for (driver.py: no ``.cpu()``, no host round trip per step).  At N > 1 only rank 0
  evaluates -- on that stream, so no rank waits at its next all-reduce.
* ``HostEvaluator`` (host envs: pz / smaclite / the CPU twins): ``num_eval_ep`` in-process env instances stepped together, ONE
  ``HostActor.act`` per time step for all of a rank's episodes; at N > 1 the episodes are dealt to the ranks in contiguous blocks
  (every rank evaluates its share, one small all_gather of the per-episode results) so that no rank idles while rank 0 plays them all.

Both draw their randomness from the counter-based generator with the keys of the training rollouts -- env ``EVAL_ENV_BASE + j`` of
evaluation round ``n`` is episode ``n`` of that env index, its actions are keyed (act_seed(n), global row, t) -- so the two
evaluators produce the SAME episodes on the synthetic envs (tests/test_eval.py) and evaluation never advances the training
sampler's counters (resume stays bit-exact).  Logged tags are the reference's: eval/ep_reward, eval/std_ep_reward, eval/ep_length
(+ eval/battle_won for smaclite).
"""
import numpy as np
import torch

from . import _native as N

EVAL_ENV_BASE = 10 ** 6  # env-index range of the evaluation episodes (the training envs are [0, batch_size))
_GOLD = 0x9E3779B97F4A7C15
_EVAL_STREAM = {}


def eval_base(batch_size):
    return max(EVAL_ENV_BASE, int(batch_size))


def act_seed(seed, episode):
    """The action-sampler seed of episode `episode` (the formula of rollout.py / gru.py's device rollouts)."""
    return (int(seed) + (int(episode) + 1) * _GOLD) & 0xFFFFFFFFFFFFFFFF


def eval_stream(device):
    """The evaluation rollouts' own lowest-priority stream (one per device and process, like _native.low_priority_stream -- a SECOND
    one: the critic's epochs of the overlapped schedules must not queue behind an evaluation episode's latency chain)."""
    key = torch.device(device).index or 0
    if key not in _EVAL_STREAM:
        h = N.load().cm_stream_create_low_priority()
        if not h:
            raise N.NativeError("cm_stream_create_low_priority failed: " + (N.load().cm_last_error() or b"?").decode())
        _EVAL_STREAM[key] = torch.cuda.ExternalStream(h, device=device)
    return _EVAL_STREAM[key]


class EvalResult:
    """Per-episode returns / lengths of one evaluation round, possibly still on their way from the device."""

    def __init__(self, rewards=None, lengths=None, won=None, pinned=None, event=None, T=None):
        self._r, self._l, self._w, self._pin, self._ev, self._T = rewards, lengths, won, pinned, event, T

    def _get(self):
        if self._r is None:
            self._ev.synchronize()
            self._r = self._pin.tolist()
            self._l = [self._T] * len(self._r)
        return self._r, self._l, self._w

    @property
    def ep_rewards(self):
        return self._get()[0]

    @property
    def ep_lengths(self):
        return self._get()[1]

    def log(self, writer, step, smaclite=False):
        r, l, w = self._get()
        writer.add_scalar("eval/ep_reward", np.mean(r), step)
        writer.add_scalar("eval/std_ep_reward", np.std(r), step)
        writer.add_scalar("eval/ep_length", np.mean(l), step)
        if smaclite:
            writer.add_scalar("eval/battle_won", np.mean(w), step)


class DeviceEvaluator:
    """num_eval_ep evaluation episodes of a device env as one rollout on the evaluation stream."""

    def __init__(self, args, actor_spec, n_agents, device, batch_size):
        from .rollout import SyntheticShapeRollout, SyntheticSpreadRollout
        self.n, self.spec, self.device = int(args.num_eval_ep), actor_spec, device
        base = eval_base(batch_size)
        T = args.synthetic_steps
        if args.env_type == "synthetic_shape":
            self.roll = SyntheticShapeRollout(self.n, n_agents, T, obs_raw=args.synthetic_obs, state_dim=args.synthetic_state,
                                              n_actions=args.synthetic_actions, avail_p=args.synthetic_avail_p, seed=args.seed,
                                              agent_ids=args.agent_ids, device=device, env_offset=base, pad=actor_spec.kind != "gru")
        elif actor_spec.kind == "gru":
            from .gru import GRUSyntheticRollout
            # --greedy_eval rollouts take the per-step act path (the fused recurrent rollout has no argmax sampler): contiguous buffers from
            # the start, so the first launch() neither synchronises the device nor allocates on the evaluation stream
            self.roll = GRUSyntheticRollout(self.n, n_agents, T, seed=args.seed, agent_ids=args.agent_ids, device=device, env_offset=base,
                                            pad_state=not getattr(args, "greedy_eval", False))
            if getattr(args, "greedy_eval", False):  # the per-step path's hidden-state and workspace buffers, allocated here, not in launch()
                self.roll.h = torch.zeros(self.n * n_agents, actor_spec.hidden, dtype=torch.float32, device=device)
        else:
            self.roll = SyntheticSpreadRollout(self.n, n_agents, T, seed=args.seed, agent_ids=args.agent_ids, device=device, env_offset=base)
        self.stream = eval_stream(device)
        self.snap = None
        self.pins = [torch.empty(self.n, dtype=torch.float32, pin_memory=True) for _ in range(2)]
        self.events = [None, None]
        self.k = 0

    def launch(self, actor_flat, round_index, greedy=False):
        """Enqueue evaluation round `round_index` with the actor's CURRENT parameters (as of the caller's stream) and return at once."""
        main = torch.cuda.current_stream(self.device)
        k = self.k = self.k ^ 1
        if self.events[k] is not None:
            self.events[k].synchronize()  # the round before last: long done (its result was logged an iteration later)
        if self.snap is None:
            self.snap = [torch.empty_like(actor_flat) for _ in range(2)]
        # The parameters are snapshotted ON THE CALLER'S STREAM (one 33 KB copy behind the update that produced them: the next optimiser
        # step may overwrite `actor_flat` at once) and the evaluation stream waits for that copy only -- the training stream never waits
        # for the low-priority stream.  Two snapshots / result slots alternate: round n - 2 was waited for above.
        snap = self.snap[k]
        snap.copy_(actor_flat, non_blocking=True)
        copied = torch.cuda.Event()
        copied.record(main)
        with torch.cuda.stream(self.stream):
            self.stream.wait_event(copied)
            self.roll.episode = int(round_index)
            b = self.roll.collect(snap, self.spec, eps=-1.0 if greedy else 0.0)
            self.pins[k].copy_(b.reward.sum(1), non_blocking=True)
            self.events[k] = torch.cuda.Event()
            self.events[k].record()
        return EvalResult(pinned=self.pins[k], event=self.events[k], T=b.T)


class HostEvaluator:
    """num_eval_ep evaluation episodes of a host env stepped side by side: one act call per time step for all of this rank's episodes.

    Host cost: a rank keeps one in-process env per episode slot it owns (the reference reuses ONE eval env sequentially,
    cleanmarl/mappo_multienvs.py:614-650).  The envs are created on first use, and ``--eval_live_envs=N`` (N > 0) bounds how many are alive
    at once: the slots are then played in waves of N.  Heavy envs whose episodes are NOT keyed by (index, episode) counters (smaclite,
    pettingzoo) are POOLED: at most N envs are ever built, wave position i of every wave and every round reuses pool env i and only
    resets it -- N = 1 is exactly the reference's sequential evaluation on one env whose RNG advances (no construction per episode, no
    re-seeding per round).  The counter-keyed CPU twins of the synthetic envs ARE their index (an episode is a function of
    (seed, index, episode)): they are cheap to build, so a wave's envs are closed before the next wave is built.  The action keys do not
    depend on the wave size (row = global slot x agent), so every setting plays the same episodes on the counter-keyed envs."""

    def __init__(self, make_env, first_env, host_actor, args, n_agents, recurrent, device, batch_size, rank=0, world=1, pg=None):
        """make_env(index) -> CommonInterface env; first_env: the script's eval_env (index eval_base), reused as episode slot 0."""
        self.n, self.A, self.recurrent, self.device = int(args.num_eval_ep), n_agents, recurrent, device
        self.actor, self.seed = host_actor, int(args.seed)
        self.base = eval_base(batch_size)
        self.rank, self.world, self.pg = rank, world, pg
        self.per = (self.n + world - 1) // world
        self.mine = list(range(min(self.n, rank * self.per), min(self.n, (rank + 1) * self.per)))  # a contiguous block of episode slots
        self.make_env, self.first_env = make_env, first_env
        self.max_live = max(0, int(getattr(args, "eval_live_envs", 0) or 0))
        self.envs = {}  # slot -> env, created on first use (_env)
        self.pool = {}  # wave position -> env of the pooled (not counter-keyed) envs when --eval_live_envs bounds them; position 0 = first_env
        self.pooled = self.max_live > 0 and not hasattr(first_env, "episode")
        self.smaclite = args.env_type == "smaclite"
        self.record, self.actions = False, None  # tests: the actions of the last round, [wave * t][slot of the wave][agent]

    def _env(self, j, pos=0):
        """The env that plays slot j; pos = its position in the wave (pooled envs are owned by the position, not by the slot)."""
        if self.pooled:
            if pos not in self.pool:
                self.pool[pos] = self.first_env if pos == 0 else self.make_env(self.base + pos)
            return self.pool[pos]
        if j not in self.envs:
            self.envs[j] = self.first_env if j == 0 else self.make_env(self.base + j)
        return self.envs[j]

    def _release(self, slots):
        for j in slots:
            e = self.envs.pop(j, None)
            if e is not None and j != 0:  # slot 0 is the caller's eval_env
                e.close()

    def close(self):
        self._release(list(self.envs))
        for pos, e in list(self.pool.items()):
            if pos != 0:  # position 0 is the caller's eval_env
                e.close()
        self.pool.clear()

    def _play(self, slots, round_index, greedy, eps, seed):
        """One wave: the episodes of `slots` (contiguous) side by side.  The act call always carries ALL of the wave's slots (finished
        episodes ride along with their last observation, their actions are dropped): row (base + j) * A + a of step t is keyed the same
        whatever the other episodes do, however the slots are dealt over ranks and waves -- exactly like the device evaluator's rollout."""
        A = self.A
        envs = {j: self._env(j, i) for i, j in enumerate(slots)}
        for j in slots:
            if hasattr(envs[j], "episode"):  # the counter-keyed CPU twins: round n IS episode n of env base + j (resume-safe)
                envs[j].episode = int(round_index) - 1
        obs = {j: envs[j].reset()[0] for j in slots}
        ret = {j: 0.0 for j in slots}; length = {j: 0 for j in slots}; info = {j: None for j in slots}
        alive = set(slots)
        h, t = None, 0
        while alive:
            x = np.stack([np.asarray(obs[j], np.float32) for j in slots])
            av = np.stack([np.asarray(envs[j].get_avail_actions()) for j in slots])
            act, _, h = self.actor.act(x, av, h=h, seed=seed, greedy=greedy, eps=eps, t=t, row_offset=(self.base + slots[0]) * A)
            act = np.asarray(act).reshape(len(slots), A)
            if self.record:
                self.actions.append(act.copy())
            for i, j in enumerate(slots):
                if j not in alive:
                    continue
                o, rew, done, trunc, inf = envs[j].step(act[i])
                ret[j] += rew; length[j] += 1
                if done or trunc:
                    info[j] = inf
                    alive.discard(j)
                else:
                    obs[j] = o
            t += 1
        won = [float(info[j]["battle_won"]) if self.smaclite else 0.0 for j in slots]
        return [ret[j] for j in slots], [float(length[j]) for j in slots], won

    def run(self, round_index, greedy=False, eps=0.0):
        """Evaluation round `round_index`: every slot of this rank plays one episode (in waves of --eval_live_envs when that is set)."""
        slots = self.mine
        seed = act_seed(self.seed, round_index)
        r, l, won = [], [], []
        self.actions = []
        step = self.max_live if self.max_live > 0 else max(1, len(slots))
        for w0 in range(0, len(slots), step):
            wave = slots[w0:w0 + step]
            rw, lw, ww = self._play(wave, round_index, greedy, eps, seed)
            r += rw; l += lw; won += ww
            if self.max_live > 0 and len(slots) > self.max_live and not self.pooled:
                self._release(wave)  # the next wave's envs take their place (pooled envs stay: they are reset, not rebuilt)
        if self.world > 1:  # every rank played its block: gather (slot order, as if one process had played them all)
            loc = torch.full((3, self.per), float("nan"), dtype=torch.float64, device=self.device)
            if slots:
                loc[0, :len(slots)] = torch.tensor(r, dtype=torch.float64); loc[1, :len(slots)] = torch.tensor(l, dtype=torch.float64)
                loc[2, :len(slots)] = torch.tensor(won, dtype=torch.float64)
            parts = [torch.empty_like(loc) for _ in range(self.world)]
            torch.distributed.all_gather(parts, loc, group=self.pg)
            allp = torch.cat(parts, 1).cpu()  # rank-major = slot order (contiguous blocks); unused positions are NaN
            keep = ~torch.isnan(allp[1])
            r, l, won = allp[0][keep].tolist(), allp[1][keep].tolist(), allp[2][keep].tolist()
        return EvalResult(rewards=r, lengths=l, won=won)
