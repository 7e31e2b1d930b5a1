"""Training driver behind coma_multienvs.py and coma.py.  Control flow follows the reference's ``__main__`` block
(cleanmarl/coma_multienvs.py:352-737; coma.py for the single-environment front-end): per iteration epsilon =
linear_schedule(training_step), rollout with the epsilon-mixed policy, targets from the target critic, one critic step,
polyak update, one actor step, logging, periodic eval (sampled with eps = 0).  All numerics run in libcleanmarl_hip.so.
Multi-GPU: env-sharded like driver.py (one all-reduce per optimiser step + the [T][4] advantage-moment sums).
"""
import datetime
import os
import random

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC only on these hosts (RCCL needs it for world > 1)

import numpy as np  # noqa: E402
import torch  # noqa: E402

from . import _native as N
from .args import parse_args
from .coma_learner import COMAHParams, COMALearner, coma_critic_input_dim
from .driver import HostActor, host_rollout, host_rollout_shm, host_rollout_single
from .env.shm_vector import ShmVectorEnv
from .env.vector import PipeVectorEnv, environment
from .evaluate import DeviceEvaluator, HostEvaluator, eval_base
from .learner import NetSpec, init_params_like_torch, pad_time
from .logger import ScalarWriter
from .rollout import SyntheticShapeRollout, SyntheticSpreadRollout

RUN_PREFIX = {"coma_multienvs": "COMA-multienvs", "coma": "COMA"}  # coma_multienvs.py:424, coma.py:348


def linear_schedule(start_e, end_e, duration, t):
    """cleanmarl/coma_multienvs.py:243-245."""
    slope = (end_e - start_e) / duration
    return max(slope * t + start_e, end_e)


def run(script, argv=None):
    args = parse_args(script, argv)
    single_env = script == "coma"
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not str(args.device).startswith("cuda"):
        raise N.NativeError(f"--device={args.device}: this build computes on MI355X only (cuda / cuda:N); there is no CPU path")
    random.seed(args.seed); np.random.seed(args.seed); torch.manual_seed(args.seed)
    # CM_DIST_BACKEND=gloo is a TEST hook (tests/test_dist_gpu.py): RCCL refuses two ranks on one device, gloo does not, so the N > 1 code
    # path of the drivers can be exercised on a 1-GPU box with every rank on cuda:0
    backend = os.environ.get("CM_DIST_BACKEND", "nccl")
    dev_index = local_rank if backend == "nccl" else local_rank % max(1, torch.cuda.device_count())
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    pg = None
    force = world == 1 and os.environ.get("CM_FORCE_COLLECTIVES") == "1"  # test hook: see driver.run
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
    E = E_glob // world + (1 if rank < E_glob % world else 0)
    env_offset = rank * (E_glob // world) + min(rank, E_glob % world)
    synth = dict(agents=args.synthetic_agents, steps=args.synthetic_steps, obs=args.synthetic_obs, state=args.synthetic_state,
                 actions=args.synthetic_actions, avail_p=args.synthetic_avail_p)
    fac = dict(env_type=args.env_type, env_name=args.env_name, env_family=args.env_family, agent_ids=args.agent_ids,
               kwargs={}, seed=args.seed, synthetic=synth)
    eval_env = environment(**dict(fac, index=eval_base(E_glob)))
    A, Do, Ds, K = eval_env.n_agents, eval_env.get_obs_size(), eval_env.get_state_size(), eval_env.get_action_size()

    # construction order actor -> critic (coma_multienvs.py:391-405); the target starts as a copy of the critic (:406)
    actor_spec = NetSpec(Do, args.actor_hidden_dim, args.actor_num_layers, K)
    critic_spec = NetSpec(coma_critic_input_dim(Do, Ds, A, K), args.critic_hidden_dim, args.critic_num_layers, K)
    a_init, c_init = init_params_like_torch(actor_spec), init_params_like_torch(critic_spec)
    learner = COMALearner(actor_spec, critic_spec, A, COMAHParams.from_args(args), device, a_init, c_init, pg, world)

    device_env = args.env_type in ("synthetic", "synthetic_shape")
    venv = roll = the_env = pinned = None
    if args.env_type == "synthetic_shape":
        roll = SyntheticShapeRollout(E, A, args.synthetic_steps, obs_raw=args.synthetic_obs, state_dim=args.synthetic_state,
                                     n_actions=args.synthetic_actions, avail_p=args.synthetic_avail_p, seed=args.seed,
                                     agent_ids=args.agent_ids, device=device, env_offset=env_offset)
    elif device_env:
        roll = SyntheticSpreadRollout(E, A, args.synthetic_steps, seed=args.seed, agent_ids=args.agent_ids, device=device,
                                      env_offset=env_offset)  # padded rows: the COMA entry points take leading dimensions (round 4)
    elif single_env:
        the_env = environment(**dict(fac, index=env_offset))
    elif args.vector_env == "pipe":
        venv = PipeVectorEnv(E, dict(fac, synthetic=synth), index_offset=env_offset)
    else:
        venv = ShmVectorEnv(E, dict(fac, synthetic=synth), n_workers=args.env_workers or None, index_offset=env_offset)
        if args.vector_env == "pinned":  # default: the workers' shared blocks are page-locked and copied straight into the device buffer
            from .host_rollout import PinnedHostRollout
            pinned = PinnedHostRollout(venv, learner, False, device, row_offset=env_offset * A, pad=False)
    host_actor = HostActor(learner, A, False, device, row_offset=env_offset * A)

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
    step = 0
    if args.checkpoint and os.path.exists(args.checkpoint):
        ck = torch.load(args.checkpoint, map_location="cpu")
        learner.load_state_dict(ck["learner"])
        step = ck["step"]
        if roll is not None:
            roll.episode = ck.get("episode", 0)

    def save_checkpoint():
        if args.checkpoint and rank == 0:
            torch.save(dict(learner=learner.state_dict(), step=step, episode=roll.episode if roll is not None else 0), args.checkpoint)

    iteration = 0
    evaluator, pending_eval = None, []
    while step < args.total_timesteps:
        training_step = learner.training_step
        epsilon = linear_schedule(args.start_e, args.end_e, args.exploration_fraction, training_step)  # :449-451
        if device_env:
            b = roll.collect(learner.actor, actor_spec, eps=epsilon)
            stats = dict(ep_reward=b.reward.sum(1).cpu().tolist(), ep_len=[b.T] * E, infos=[None] * E)
        elif single_env:
            b, stats = host_rollout_single(the_env, host_actor, E, A, args.seed + training_step, False, device, explore=epsilon)
        else:
            if pinned is not None:
                b, stats = pinned.collect(args.seed + training_step, eps=epsilon)
            else:
                collect = host_rollout if args.vector_env == "pipe" else host_rollout_shm
                b, stats = collect(venv, host_actor, E, A, args.seed + training_step, False, device, explore=epsilon)
        n_steps = torch.tensor([float(sum(stats["ep_len"]))], device=device)
        if world > 1:
            torch.distributed.all_reduce(n_steps, group=pg)
            if not device_env:
                t_max = torch.tensor([b.T], device=device)
                torch.distributed.all_reduce(t_max, op=torch.distributed.ReduceOp.MAX, group=pg)
                b = pad_time(b, int(t_max.item()))
        step += int(n_steps.item())
        ep_rewards.extend(stats["ep_reward"]); ep_lengths.extend(stats["ep_len"])
        if args.env_type == "smaclite":
            ep_stats.extend([i["battle_won"] for i in stats["infos"]])
        # rollout logging cadence and the (sic) episode counters of the two scripts: coma_multienvs.py:530-545, coma.py:419-434
        log_now = (len(ep_rewards) > args.log_every) if single_env else (training_step % args.log_every == 0)
        if log_now:
            if writer:
                writer.add_scalar("rollout/ep_reward", np.mean(ep_rewards), step)
                writer.add_scalar("rollout/ep_length", np.mean(ep_lengths), step)
                writer.add_scalar("rollout/epsilon", epsilon, step)
                writer.add_scalar("rollout/num_episodes", ((training_step + 1) if single_env else training_step) * E_glob, step)
                if args.env_type == "smaclite":
                    writer.add_scalar("rollout/battle_won", np.mean(ep_stats), step)
            ep_rewards, ep_lengths, ep_stats = [], [], []

        rec = learner.train_iteration(b)
        iteration += 1
        if args.checkpoint_every and iteration % args.checkpoint_every == 0:
            save_checkpoint()
        if writer:  # :678-684
            writer.add_scalar("train/critic_loss", rec["critic_loss"], step)
            writer.add_scalar("train/actor_loss", rec["actor_loss"], step)
            writer.add_scalar("train/entropy", rec["entropy"], step)
            writer.add_scalar("train/actor_gradients", rec["actor_gnorm"], step)
            writer.add_scalar("train/critic_gradients", rec["critic_gnorm"], step)
            writer.add_scalar("train/epsilon", epsilon, step)
            writer.add_scalar("train/num_updates", learner.training_step, step)

        for log_eval in pending_eval:  # the evaluation launched behind the previous iteration ran under this one (evaluate.py)
            log_eval()
        pending_eval.clear()
        if learner.training_step % args.eval_steps == 0:  # :692-727 (actions sampled with eps = 0), num_eval_ep episodes side by side
            eval_round = learner.training_step // args.eval_steps
            greedy = bool(getattr(args, "greedy_eval", False))
            if device_env:
                if rank == 0:
                    if evaluator is None:
                        evaluator = DeviceEvaluator(args, actor_spec, A, device, E_glob)
                    res = evaluator.launch(learner.actor, eval_round, greedy=greedy)
                    pending_eval.append(lambda res=res, step=step: res.log(writer, step))
            else:
                if evaluator is None:
                    evaluator = HostEvaluator(lambda index: environment(**dict(fac, index=index)), eval_env, host_actor, args, A, False,
                                              device, E_glob, rank, world, pg)
                res = evaluator.run(eval_round, greedy=greedy)
                if rank == 0:
                    res.log(writer, step, smaclite=args.env_type == "smaclite")

    for log_eval in pending_eval:
        log_eval()
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
    if world > 1 or force:
        torch.distributed.destroy_process_group()
    return dict(step=step, training_step=learner.training_step, history=writer.history if writer else [], learner=learner)
