"""Training driver behind the single-file scripts: the vectorised mappo_multienvs.py, ippo_multienvs.py and their
*_lstm_* pair, plus the single-environment front-ends mappo.py, ippo.py, mappo_lstm.py, ippo_lstm.py (same learner,
episodes collected one after the other from one in-process env).  Control flow follows the reference's ``__main__`` block (cleanmarl/mappo_multienvs.py:288-659):
seed -> envs -> networks/optimisers -> writer -> while step < total_timesteps: rollout, TD(lambda), epochs of
PPO, logging, periodic eval -> shutdown.  All numerics run in libcleanmarl_hip.so on the GPU.

Multi-GPU: launched one process per GPU (torchrun); ``--batch_size`` environments are sharded over ranks and
the only data-path collective is the per-optimiser-step all-reduce of the flat gradient buffer (learner.py).
"""
import datetime
import os
import random

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC only on these hosts (RCCL needs it for world > 1)

import numpy as np  # noqa: E402
import torch  # noqa: E402

from . import _native as N
from .args import parse_args
from .env.shm_vector import ShmVectorEnv
from .env.vector import PipeVectorEnv, environment
from .evaluate import DeviceEvaluator, HostEvaluator, eval_base
from .learner import DeviceBatch, HParams, NetSpec, PPOLearner, init_params_like_torch, pad_time
from .logger import ScalarWriter
from .rollout import SyntheticShapeRollout, SyntheticSpreadRollout

RUN_PREFIX = {  # run-name strings of the four scripts (sic)
    "mappo_multienvs": "MAPPO-multienvs", "ippo_multienvs": "IPPO-multienvs",
    "mappo_lstm_multienvs": "MAPPO-lstm-multienv", "ippo_lstm_multienvs": "IPPO-lstm-multienvs",
    # single-environment front-ends (cleanmarl/mappo.py:279, ippo.py:279, mappo_lstm.py:280, ippo_lstm.py:279)
    "mappo": "MAPPO", "ippo": "IPPO", "mappo_lstm": "MAPPO-lstm", "ippo_lstm": "IPPO-lstm",
}


class HostActor:
    """Actor.act for observations that live on the host (real envs, eval): stage -> cm_policy_act -> fetch."""

    def __init__(self, learner, n_agents, recurrent, device, row_offset=0):
        """row_offset: global index of this rank's first (env, agent) row (= env_offset * A when the batch is env-sharded over
        ranks): it enters the Philox key of the sampler, so ranks draw different uniforms for their shards."""
        self.L, self.A, self.recurrent, self.dev = learner, n_agents, recurrent, device
        self.row_offset = int(row_offset)
        self.lib = N.load()
        self.calls = 0
        self.ws = None

    def act(self, obs, avail, h=None, seed=0, greedy=False, eps=0.0, t=None, row_offset=None):
        """t / row_offset: explicit Philox step counter / global index of the first row (the evaluators key their draws like the device
        rollouts: evaluate.py); default: this actor's own call counter and row offset (the training rollouts of the host envs)."""
        spec = self.L.actor_spec
        x = torch.as_tensor(np.ascontiguousarray(obs), dtype=torch.float32).reshape(-1, spec.din).to(self.dev)
        av = torch.as_tensor(np.ascontiguousarray(avail)).reshape(-1, spec.dout).to(torch.uint8).to(self.dev)
        rows = x.shape[0]
        action = torch.empty(rows, dtype=torch.int32, device=self.dev)
        logp = torch.empty(rows, dtype=torch.float32, device=self.dev)
        if t is None:
            self.calls += 1
            t = self.calls
        row_offset = self.row_offset if row_offset is None else int(row_offset)
        if self.recurrent:
            if h is None:
                h = torch.zeros(rows, spec.hidden, dtype=torch.float32, device=self.dev)
            # eps < 0: argmax (greedy evaluation); obs wider than 64 columns / more than 64 hidden units: the layered schedule's workspace
            need = self.lib.cm_gru_policy_act_workspace_bytes(rows, spec.din, spec.hidden, spec.dout)
            if self.ws is None or self.ws.numel() < need:
                self.ws = torch.empty(need, dtype=torch.uint8, device=self.dev)
            N.check(self.lib.cm_gru_policy_act_ws(N.ptr(x), spec.din, N.ptr(av), spec.dout, rows, spec.din, spec.hidden, spec.dout,
                                                  N.ptr(self.L.actor), N.ptr(h), -1.0 if greedy else 0.0, seed, row_offset, t,
                                                  N.ptr(action), N.ptr(logp), 1, N.ptr(self.ws), self.ws.numel(), N.stream_ptr()),
                    "cm_gru_policy_act_ws")
        else:
            # eps < 0: argmax of the masked logits (build option; the reference always samples); eps > 0: COMA exploration
            mode = -1.0 if greedy else float(eps)
            need = self.lib.cm_policy_act_workspace_bytes(rows, spec.din, spec.hidden, spec.n_layers, spec.dout)  # 0 unless layered
            if need and (self.ws is None or self.ws.numel() < need):
                self.ws = torch.empty(need, dtype=torch.uint8, device=self.dev)
            N.check(self.lib.cm_policy_act_ws(N.ptr(x), spec.din, N.ptr(av), spec.dout, rows, spec.din, spec.hidden, spec.n_layers,
                                              spec.dout, N.ptr(self.L.actor), mode, seed, row_offset, t, N.ptr(action), N.ptr(logp), 1,
                                              N.ptr(self.ws) if need else None, need, N.stream_ptr()), "cm_policy_act_ws")
        return action.cpu().numpy(), logp.cpu().numpy(), h


def host_rollout(venv, actor, E, A, seed, recurrent, device, explore=0.0, pad=False):
    """One episode per env over the pipe protocol (cleanmarl/mappo_multienvs.py:393-453; GRU: hidden state
    carried for alive envs only, cleanmarl/mappo_lstm_multienvs.py:406-433).  Returns (DeviceBatch, stats)."""
    cont = venv.reset_all()
    obs = np.stack([c["obs"] for c in cont]); avail = np.stack([c["avail_actions"] for c in cont])
    state = np.stack([c["state"] for c in cont])
    eps = [dict(obs=[], actions=[], logp=[], reward=[], state=[], avail=[]) for _ in range(E)]
    alive = list(range(E))
    ep_reward, ep_len, ep_info = [0.0] * E, [0] * E, [None] * E
    h_all = None
    while alive:
        h_in = None
        if recurrent and h_all is not None:
            idx = torch.as_tensor(alive, device=device)
            h_in = h_all.reshape(E, A, -1)[idx].reshape(len(alive) * A, -1).contiguous()
        act, logp, h_out = actor.act(obs, avail, h=h_in, seed=seed, eps=explore)
        if recurrent:
            if h_all is None:
                h_all = h_out
            else:
                h_all.reshape(E, A, -1)[torch.as_tensor(alive, device=device)] = h_out.reshape(len(alive), A, -1)
        act = act.reshape(len(alive), A); logp = logp.reshape(len(alive), A)
        cont = venv.step(alive, [list(map(int, a)) for a in act])
        nobs, nstate, navail, still = [], [], [], []
        for i, j in enumerate(alive):
            e, c = eps[j], cont[i]
            e["obs"].append(obs[i]); e["actions"].append(act[i]); e["logp"].append(logp[i]); e["reward"].append(c["reward"])
            e["state"].append(state[i]); e["avail"].append(avail[i])
            ep_reward[j] += c["reward"]; ep_len[j] += 1
            if c["done"] or c["truncated"]:
                ep_info[j] = c.get("infos")
            else:
                still.append(j); nobs.append(c["next_obs"]); nstate.append(c["next_state"]); navail.append(c["avail_actions"])
        alive = still
        if alive:
            obs, state, avail = np.stack(nobs), np.stack(nstate), np.stack(navail)
    return _collate(eps, ep_len, E, A, device, pad), dict(ep_reward=ep_reward, ep_len=ep_len, infos=ep_info)


def _collate(eps, ep_len, E, A, device, pad=False):
    """Zero-pad the per-episode lists to the longest episode + build the mask (RolloutBuffer.get_batch, :109-142)."""
    T = max(ep_len)
    Do, Ds, K = eps[0]["obs"][0].shape[-1], eps[0]["state"][0].shape[-1], eps[0]["avail"][0].shape[-1]
    b_obs = np.zeros((E, T, A, Do), np.float32); b_av = np.zeros((E, T, A, K), bool); b_act = np.zeros((E, T, A), np.int64)
    b_lp = np.zeros((E, T, A), np.float32); b_rew = np.zeros((E, T), np.float32); b_st = np.zeros((E, T, Ds), np.float32)
    b_mask = np.zeros((E, T), bool)
    for j, e in enumerate(eps):
        n = ep_len[j]
        b_obs[j, :n] = np.stack(e["obs"]); b_av[j, :n] = np.stack(e["avail"]).astype(bool); b_act[j, :n] = np.stack(e["actions"])
        b_lp[j, :n] = np.stack(e["logp"]); b_rew[j, :n] = np.asarray(e["reward"], np.float32); b_st[j, :n] = np.stack(e["state"])
        b_mask[j, :n] = True
    t = torch.from_numpy
    return DeviceBatch.from_reference_layout(t(b_obs), t(b_act), t(b_lp), t(b_rew), t(b_st), t(b_av), t(b_mask), device, pad=pad)


def host_rollout_single(env, actor, E, A, seed, recurrent, device, explore=0.0, pad=False):
    """The single-environment front-ends (cleanmarl/mappo.py:302-343, ippo.py, mappo_lstm.py, ippo_lstm.py): ``batch_size``
    episodes collected ONE AFTER THE OTHER from one in-process env (no worker processes); the GRU hidden state starts at
    zero with every episode (mappo_lstm.py:306-318).  Returns the same (DeviceBatch, stats) as the vectorised collectors."""
    eps, ep_reward, ep_len, ep_info = [], [], [], []
    for _ in range(E):
        e = dict(obs=[], actions=[], logp=[], reward=[], state=[], avail=[])
        obs, _ = env.reset()
        done = trunc = False
        tot, n, h, info = 0.0, 0, None, None
        while not (done or trunc):
            avail, state = np.asarray(env.get_avail_actions()), np.asarray(env.get_state())
            act, logp, h = actor.act(np.asarray(obs)[None], avail[None], h=h, seed=seed, eps=explore)
            nobs, r, done, trunc, info = env.step(act.reshape(-1))
            e["obs"].append(np.asarray(obs, np.float32)); e["actions"].append(act.reshape(A)); e["logp"].append(logp.reshape(A))
            e["reward"].append(r); e["state"].append(state); e["avail"].append(avail)
            tot += r; n += 1
            obs = nobs
        eps.append(e); ep_reward.append(tot); ep_len.append(n); ep_info.append(info)
    return _collate(eps, ep_len, E, A, device, pad), dict(ep_reward=ep_reward, ep_len=ep_len, infos=ep_info)


def host_rollout_shm(venv, actor, E, A, seed, recurrent, device, explore=0.0, pad=False):
    """Same episode collection through the shared-memory batched-step vector env: one token per
    WORKER per step instead of one pickled round trip per ENV.  Returns the same (DeviceBatch, stats)."""
    hstate = {"h": None}

    def act_fn(obs, avail, alive):
        h_in = None
        if recurrent and hstate["h"] is not None:
            idx = torch.as_tensor(alive, device=device)
            h_in = hstate["h"].reshape(E, A, -1)[idx].reshape(len(alive) * A, -1).contiguous()
        act, logp, h_out = actor.act(obs, avail, h=h_in, seed=seed, eps=explore)
        if recurrent:
            if hstate["h"] is None:
                hstate["h"] = h_out
            else:
                hstate["h"].reshape(E, A, -1)[torch.as_tensor(alive, device=device)] = h_out.reshape(len(alive), A, -1)
        return act.reshape(len(alive), A), logp.reshape(len(alive), A)

    out, mask, stats = venv.collect_episode(act_fn)
    t = torch.from_numpy
    b = DeviceBatch.from_reference_layout(t(out["obs"]), t(out["act"]), t(out["logp"]), t(out["rew"]), t(out["state"]),
                                          t(out["avail"]), t(mask), device, pad=pad)
    return b, stats


def run(script, argv=None):
    args = parse_args(script, argv)
    algo = "mappo" if script.startswith("mappo") else "ippo"
    recurrent = "lstm" in script
    single_env = not script.endswith("multienvs")  # mappo.py / ippo.py / mappo_lstm.py / ippo_lstm.py
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not str(args.device).startswith("cuda"):
        raise N.NativeError(f"--device={args.device}: this build computes on MI355X only (cuda / cuda:N); there is no CPU path")
    random.seed(args.seed); np.random.seed(args.seed); torch.manual_seed(args.seed)  # :291-294
    # CM_DIST_BACKEND=gloo is a TEST hook (tests/test_dist_gpu.py): RCCL refuses two ranks on one device, gloo does not, so the N > 1 code
    # path of the drivers can be exercised on a 1-GPU box with every rank on cuda:0
    backend = os.environ.get("CM_DIST_BACKEND", "nccl")
    dev_index = local_rank if backend == "nccl" else local_rank % max(1, torch.cuda.device_count())
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    pg = None
    # CM_FORCE_COLLECTIVES=1 (test hook): a ONE-rank process group, so that the learners issue every collective of an env-sharded run --
    # over RCCL itself on a one-GPU box (a one-rank sum is the identity: the run must equal the plain one)
    force = world == 1 and os.environ.get("CM_FORCE_COLLECTIVES") == "1"
    if force:
        os.environ.setdefault("MASTER_PORT", "29531")
    if world > 1 or force:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            torch.distributed.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
        else:
            torch.distributed.init_process_group(backend, rank=rank, world_size=world)
        pg = torch.distributed.group.WORLD
    E_glob = args.batch_size
    if E_glob < world:
        raise N.NativeError(f"--batch_size={E_glob} < {world} ranks: every rank needs at least one environment (env-sharded data parallelism)")
    E = E_glob // world + (1 if rank < E_glob % world else 0)  # env shard of this rank
    env_offset = rank * (E_glob // world) + min(rank, E_glob % world)
    synth = dict(agents=args.synthetic_agents, steps=args.synthetic_steps, obs=args.synthetic_obs, state=args.synthetic_state,
                 actions=args.synthetic_actions, avail_p=args.synthetic_avail_p)
    fac = dict(env_type=args.env_type, env_name=args.env_name, env_family=args.env_family, agent_ids=args.agent_ids,
               kwargs={}, seed=args.seed, synthetic=synth)
    eval_env = environment(**dict(fac, index=eval_base(E_glob)))
    A, Do, Ds, K = eval_env.n_agents, eval_env.get_obs_size(), eval_env.get_state_size(), eval_env.get_action_size()

    # networks in the reference's construction order actor -> critic (:329-339) so torch.manual_seed reproduces them
    actor_spec = NetSpec(Do, args.actor_hidden_dim, 0 if recurrent else args.actor_num_layers, K, "gru" if recurrent else "mlp")
    critic_spec = NetSpec(Ds if algo == "mappo" else Do, args.critic_hidden_dim, args.critic_num_layers, 1)
    a_init, c_init = init_params_like_torch(actor_spec), init_params_like_torch(critic_spec)
    hp = HParams.from_args(args)
    if recurrent:
        from .gru import GRUPPOLearner
        learner = GRUPPOLearner(algo, actor_spec, critic_spec, A, hp, device, a_init, c_init, pg, world)
    else:
        learner = PPOLearner(algo, actor_spec, critic_spec, A, hp, device, a_init, c_init, pg, world)
    learner.global_envs = E_glob  # the update schedule is chosen from the largest shard, identically on every rank (learner._schedule_rows)

    device_env = args.env_type in ("synthetic", "synthetic_shape")
    venv = roll = the_env = pinned = None
    if args.env_type == "synthetic_shape":
        roll = SyntheticShapeRollout(E, A, args.synthetic_steps, obs_raw=args.synthetic_obs, state_dim=args.synthetic_state,
                                     n_actions=args.synthetic_actions, avail_p=args.synthetic_avail_p, seed=args.seed,
                                     agent_ids=args.agent_ids, device=device, env_offset=env_offset, pad=not recurrent)
    elif device_env:
        if recurrent:
            from .gru import GRUSyntheticRollout
            roll = GRUSyntheticRollout(E, A, args.synthetic_steps, seed=args.seed, agent_ids=args.agent_ids, device=device,
                                       env_offset=env_offset)
        else:
            roll = SyntheticSpreadRollout(E, A, args.synthetic_steps, seed=args.seed, agent_ids=args.agent_ids, device=device,
                                          env_offset=env_offset)
    elif single_env:
        the_env = environment(**dict(fac, index=env_offset))  # cleanmarl/mappo.py:235-241: one in-process env
    elif args.vector_env == "pipe":
        venv = PipeVectorEnv(E, dict(fac, synthetic=synth), index_offset=env_offset)
    else:
        venv = ShmVectorEnv(E, dict(fac, synthetic=synth), n_workers=args.env_workers or None, index_offset=env_offset)
        if args.vector_env == "pinned":  # default: the workers' shared blocks are page-locked and copied straight into the device buffer
            from .host_rollout import PinnedHostRollout
            pinned = PinnedHostRollout(venv, learner, recurrent, device, row_offset=env_offset * A)
    host_actor = HostActor(learner, A, recurrent, device, row_offset=env_offset * A)

    time_token = datetime.datetime.now().strftime("%Y-%m-%d_%H-%M-%S")
    run_name = f"{RUN_PREFIX[script]}-{args.env_type}__{args.env_name}__{time_token}"
    writer = None
    if rank == 0:
        if args.use_wnb:
            import wandb
            wandb.init(project=args.wnb_project, entity=args.wnb_entity, sync_tensorboard=True, config=vars(args), name=run_name)
        writer = ScalarWriter(f"runs/{run_name}")
        writer.add_text("hyperparameters", "|param|value|\n|-|-|\n%s" % "\n".join(f"|{k}|{v}|" for k, v in vars(args).items()))

    ep_rewards, ep_lengths, ep_stats = [], [], []
    training_step = num_episodes = step = 0
    if args.checkpoint and os.path.exists(args.checkpoint):  # resume (every rank loads the same replicated state)
        ck = torch.load(args.checkpoint, map_location="cpu")
        if ck.get("script", script) != script:
            raise N.NativeError(f"checkpoint {args.checkpoint} was written by {ck['script']}, not {script}")
        learner.load_state_dict(ck["learner"])
        training_step, num_episodes, step = ck["training_step"], ck["num_episodes"], ck["step"]
        if roll is not None:
            roll.episode = ck.get("episode", 0)
        host_actor.calls = ck.get("host_actor_calls", 0)  # the sampler's Philox counter of the host-env paths
        if pinned is not None:
            pinned.calls = ck.get("pinned_calls", 0)

    def save_checkpoint():
        if args.checkpoint and rank == 0:  # written next to the target and renamed: a crash mid-write never leaves a truncated file
            tmp = args.checkpoint + ".tmp"
            torch.save(dict(version=2, script=script, learner=learner.state_dict(), training_step=training_step, num_episodes=num_episodes,
                            step=step, episode=roll.episode if roll is not None else 0,
                            host_actor_calls=host_actor.calls, pinned_calls=pinned.calls if pinned is not None else 0), tmp)
            os.replace(tmp, args.checkpoint)
    iteration = 0
    # Accounting (episode statistics, rollout/* and train/* scalars) of an iteration needs numbers that come off the device.  With a
    # device env nothing on the host depends on them, so iteration i is accounted for AFTER iteration i + 1 has been enqueued: the host
    # never waits for work it has just launched and the launch queue stays fed (the CLI's steady state was 1.75 ms per iteration at 512
    # envs against 1.52 ms of bench.py's loop, profiles/r03_cli_steady_state.txt).  The accounting runs in iteration order with that
    # iteration's step / training_step (the rollout-logging cadence of mappo_lstm_multienvs from the PRE-update counter, as the reference
    # tests it), so the logged history is exactly the undeferred one; evaluation, checkpoints and the end of
    # the run flush it first.  Host envs account immediately (their rollouts wait for the host anyway).
    pending = []
    rew_ring = [None, None]
    evaluator = None  # built at the first evaluation (evaluate.py)

    def flush():
        while pending:
            pending.pop(0)()

    def make_account(stats_fn, recs, step, training_step, num_episodes, ts_rollout):
        """ts_rollout: training_step BEFORE this iteration's update -- the reference tests its rollout-logging cadence between rollout
        and update (mappo_lstm_multienvs.py:486); training_step (after the update) is what train/num_updates logs (:612)."""
        def account():
            nonlocal ep_rewards, ep_lengths, ep_stats
            stats = stats_fn()
            if world > 1:
                # rollout/* scalars and their cadence must be those of ONE process owning all batch_size envs: gather every rank's episodes
                # (variable counts per rank: pad to the largest shard) instead of logging rank 0's shard
                won = [float(i["battle_won"]) for i in stats["infos"]] if args.env_type == "smaclite" else [0.0] * E
                loc = torch.full((3, (E_glob + world - 1) // world), float("nan"), dtype=torch.float64, device=device)
                loc[0, :E] = torch.tensor(stats["ep_reward"], dtype=torch.float64); loc[1, :E] = torch.tensor(stats["ep_len"], dtype=torch.float64)
                loc[2, :E] = torch.tensor(won, dtype=torch.float64)
                parts = [torch.empty_like(loc) for _ in range(world)]
                torch.distributed.all_gather(parts, loc, group=pg)
                allp = torch.cat(parts, dim=1).cpu()
                keep = ~torch.isnan(allp[1])
                ep_rewards.extend(allp[0][keep].tolist()); ep_lengths.extend(allp[1][keep].tolist())
                if args.env_type == "smaclite":
                    ep_stats.extend(allp[2][keep].tolist())
            else:
                ep_rewards.extend(stats["ep_reward"]); ep_lengths.extend(stats["ep_len"])
                if args.env_type == "smaclite":
                    ep_stats.extend([i["battle_won"] for i in stats["infos"]])
            log_now = (ts_rollout % args.log_every == 0) if script == "mappo_lstm_multienvs" else (len(ep_rewards) > args.log_every)
            if log_now:
                if writer:
                    writer.add_scalar("rollout/ep_reward", np.mean(ep_rewards), step)
                    writer.add_scalar("rollout/ep_length", np.mean(ep_lengths), step)
                    writer.add_scalar("rollout/num_episodes", num_episodes, step)
                    if args.env_type == "smaclite":
                        writer.add_scalar("rollout/battle_won", np.mean(ep_stats), step)
                ep_rewards, ep_lengths, ep_stats = [], [], []
            if writer:  # train/* are means over epochs (:605-612)
                m = lambda k: float(np.mean([r[k] for r in recs]))
                writer.add_scalar("train/critic_loss", m("critic_loss"), step)
                writer.add_scalar("train/actor_loss", m("actor_loss"), step)
                writer.add_scalar("train/entropy", m("entropy"), step)
                writer.add_scalar("train/kl_divergence", m("kl"), step)
                writer.add_scalar("train/clipped_ratios", m("clipfrac"), step)
                writer.add_scalar("train/actor_gradients", m("actor_gnorm"), step)
                writer.add_scalar("train/critic_gradients", m("critic_gnorm"), step)
                writer.add_scalar("train/num_updates", training_step, step)
        return account

    while step < args.total_timesteps:
        if not device_env:
            # host rollouts allocate a fresh batch per iteration: the previous one (still read by the critic epochs on their own
            # stream, learner.update) must be finished with before its memory goes back to the allocator
            learner.wait_critic()
        if device_env:
            b = roll.collect(learner.actor, actor_spec)
            n_total = E_glob * b.T  # every env of a device rollout runs exactly T steps: no collective, no host wait
        elif single_env:
            b, stats = host_rollout_single(the_env, host_actor, E, A, args.seed + training_step, recurrent, device, pad=not recurrent)
        else:
            if pinned is not None:
                b, stats = pinned.collect(args.seed + training_step)
            else:
                collect = host_rollout if args.vector_env == "pipe" else host_rollout_shm
                b, stats = collect(venv, host_actor, E, A, args.seed + training_step, recurrent, device, pad=not recurrent)
        if not device_env:
            stats_fn = lambda stats=stats: stats
            n_steps = torch.tensor([float(sum(stats["ep_len"]))], device=device)
            if world > 1:
                torch.distributed.all_reduce(n_steps, group=pg)
                # host envs end at different steps on different ranks: agree on the padded length
                t_max = torch.tensor([b.T], device=device)
                torch.distributed.all_reduce(t_max, op=torch.distributed.ReduceOp.MAX, group=pg)
                b = pad_time(b, int(t_max.item()))
            n_total = int(n_steps.item())
        step += n_total  # counts ENV steps, like the reference (:435)
        num_episodes += E_glob

        recs = learner.train_iteration(b)
        if device_env:
            # episode returns: summed on the device, copied to page-locked memory asynchronously, read when the iteration is accounted for.
            # Enqueued AFTER the update and on the stream its statistics leave on (the critic's, when the critic epochs overlap the next
            # rollout: learner.stats_stream) -- between the rollout and the value pass these two small launches sat on the launch
            # stream's dependent chain (23 us of the 512-env share's 1.39 ms in the kernel trace of the CLI).  The rollout buffers
            # alternate and this iteration is accounted for (host wait on the event below) before the rollout after next is enqueued.
            k = iteration & 1
            if rew_ring[k] is None or rew_ring[k][0].numel() != E:
                rew_ring[k] = [torch.empty(E, dtype=torch.float32, pin_memory=True), None]
            acct_stream = getattr(learner, "stats_stream", None) or torch.cuda.current_stream(device)
            with torch.cuda.stream(acct_stream):
                rew_ring[k][0].copy_(b.reward.sum(1), non_blocking=True)
                rew_ring[k][1] = torch.cuda.Event()
                rew_ring[k][1].record()

            def stats_fn(slot=rew_ring[k], T_=b.T):
                slot[1].synchronize()
                return dict(ep_reward=slot[0].tolist(), ep_len=[T_] * E, infos=[None] * E)
        ts_rollout = training_step
        training_step += len(recs)
        iteration += 1
        acct = make_account(stats_fn, recs, step, training_step, num_episodes, ts_rollout)
        if device_env:
            flush()                 # iteration i - 1, now that iteration i is enqueued
            pending.append(acct)
        else:
            acct()
        if args.checkpoint_every and iteration % args.checkpoint_every == 0:
            flush()
            save_checkpoint()

        if (training_step / args.epochs) % args.eval_steps == 0:  # :614-650 (actions are SAMPLED unless --greedy_eval)
            # num_eval_ep episodes side by side (evaluate.py): device envs as ONE rollout on the lowest-priority evaluation stream of rank 0
            # (it runs under the next iteration; its numbers are logged when this iteration is accounted for), host envs stepped together with
            # one act call per time step, the episodes dealt over the ranks
            eval_round = int(training_step / args.epochs) // args.eval_steps
            if device_env:
                if rank == 0:
                    if evaluator is None:
                        evaluator = DeviceEvaluator(args, actor_spec, A, device, E_glob)
                    res = evaluator.launch(learner.actor, eval_round, greedy=args.greedy_eval)
                    pending.append(lambda res=res, step=step: res.log(writer, step))
            else:
                if evaluator is None:
                    evaluator = HostEvaluator(lambda index: environment(**dict(fac, index=index)), eval_env, host_actor, args, A, recurrent,
                                              device, E_glob, rank, world, pg)
                res = evaluator.run(eval_round, greedy=args.greedy_eval)
                if rank == 0:
                    res.log(writer, step, smaclite=args.env_type == "smaclite")

    flush()
    save_checkpoint()
    if writer:
        writer.close()
    if args.use_wnb and rank == 0:
        import wandb
        wandb.finish()
    if isinstance(evaluator, HostEvaluator):
        evaluator.close()
    eval_env.close()
    if pinned is not None:
        pinned.close()
    if venv:
        venv.close()
    if the_env is not None:
        the_env.close()
    if hasattr(learner, "close"):
        learner.close()
    if world > 1 or force:
        torch.distributed.destroy_process_group()
    return dict(step=step, training_step=training_step, history=writer.history if writer else [], learner=learner)
