#!/usr/bin/env python3
"""bench.py -- MAPPO hot path on MI355X: rollout buffer -> TD(lambda) scan -> PPO update.

A "step" is ONE full training iteration of the reference's outer loop (cleanmarl/mappo_multienvs.py:379-612)
on the synthetic fixed-shape MPE-like env: rollout of E envs x A agents x T steps into the device-resident
buffer, the value pass + TD(lambda) scan, and `epochs` full-batch PPO updates (actor + critic fwd/bwd,
grad all-reduce when N > 1, fused norm/clip/Adam).  Inputs never leave HBM.

    python bench.py --gpus 1 --steps 5 --warmup 2
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Prints ONE JSON line on rank 0 (contract in the task statement): `value` = agents*envs*steps per second
summed over all ranks (weak scaling: every rank owns `--envs` environments, global env index = rank*E + e).
"""
import argparse
import json
import os
import sys
import time

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC only on these hosts (RCCL needs it for N > 1)

import torch  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOADS = {
    # name: (envs per GPU, A, T, algo, env, actor, description) -- BASELINE.json configs[2] is the headline (the metric is
    # quoted on it) and the default; the others are parity-test cases that can be timed on request (--workload)
    "cfg3": (4096, 8, 128, "mappo", "spread", "mlp", "MAPPO synthetic-MPE 4096 envs x 8 agents x 128 steps, 2x64 MLP (BASELINE.json configs[2])"),
    "cfg2": (1024, 3, 128, "mappo", "spread", "mlp", "MAPPO synthetic-MPE 1024 envs x 3 agents x 128 steps, 2x64 MLP (BASELINE.json configs[1])"),
    "cfg4": (256, 10, 256, "ippo", "shape", "mlp", "IPPO smaclite-shape synthetic, 256 envs per GPU (2048 / 8 GPUs) x 10 agents x 256 steps, 2x64 MLP (BASELINE.json configs[3])"),
    "cfg5": (1024, 5, 128, "mappo", "spread", "gru", "MAPPO-GRU synthetic-MPE 1024 envs x 5 agents x 128 steps, GRU hidden 64, tbptt 10 (BASELINE.json configs[4])"),
}
PEAK_F32_MFMA_TFLOPS = 157.3  # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, 256 CU x 4 SIMD x 64 FLOP/clk x 2.4 GHz


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="cfg3", choices=sorted(WORKLOADS))
    ap.add_argument("--envs", type=int, default=0, help="override envs per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-envs", type=int, default=64, help="envs in the bounded CPU-baseline sample")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} needs torch.distributed.run with WORLD_SIZE={args.gpus} (got {world})")
    # CM_BENCH_BACKEND=gloo is a TEST hook (tests/test_dist_gpu.py): RCCL refuses two ranks on one device, gloo does
    # not, so the N > 1 code path of this file can be exercised on a 1-GPU box with every rank on cuda:0.
    backend = os.environ.get("CM_BENCH_BACKEND", "nccl")
    ndev = torch.cuda.device_count()
    if backend == "nccl" and local_rank >= ndev:
        raise SystemExit(f"rank {rank}: LOCAL_RANK {local_rank} but only {ndev} GPU(s) visible")
    dev_index = local_rank if backend == "nccl" else local_rank % max(1, ndev)
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    pg = None
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            torch.distributed.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            torch.distributed.init_process_group(backend, rank=rank, world_size=world)
        pg = torch.distributed.group.WORLD

    from cleanmarl_amd import _native as N
    from cleanmarl_amd.gru import GRUPPOLearner, GRUSyntheticRollout
    from cleanmarl_amd.learner import HParams, NetSpec, PPOLearner, init_params_like_torch
    from cleanmarl_amd.rollout import SyntheticShapeRollout, SyntheticSpreadRollout

    E, A, T, algo, env_kind, actor_kind, desc = WORKLOADS[args.workload]
    if args.envs:
        E = args.envs
    hp = HParams()  # reference defaults: gamma .99, lambda .95, eps .2, c_ent 1e-3, lr 8e-4, epochs 3 (tbptt 10)
    if env_kind == "shape":
        roll = SyntheticShapeRollout(E, A, T, seed=1, agent_ids=True, device=dev, env_offset=rank * E)
    elif actor_kind == "gru":
        roll = GRUSyntheticRollout(E, A, T, seed=1, agent_ids=True, device=dev, env_offset=rank * E)
    else:
        roll = SyntheticSpreadRollout(E, A, T, seed=1, agent_ids=True, device=dev, env_offset=rank * E)
    aspec = NetSpec(roll.Do, 64, 0 if actor_kind == "gru" else 1, roll.K, actor_kind)
    cspec = NetSpec(roll.Ds if algo == "mappo" else roll.Do, 64, 1, 1)
    torch.manual_seed(1)  # reference construction order actor -> critic (:329-339); identical on every rank
    a_init = init_params_like_torch(aspec)
    c_init = init_params_like_torch(cspec)
    learner = (GRUPPOLearner if actor_kind == "gru" else PPOLearner)(algo, aspec, cspec, A, hp, dev, a_init, c_init, pg, world)

    def barrier():
        if world > 1:
            torch.distributed.barrier(group=pg)

    def one_step(evts=None):
        if evts is not None:
            e = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
            e[0].record()
        b = roll.collect(learner.actor, aspec)
        if evts is not None:
            e[1].record()
        learner.compute_targets(b)
        if evts is not None:
            e[2].record()
        learner.update(b)
        if evts is not None:
            e[3].record()
            evts.append(e)

    # setup pass (not one of the W warmup steps, never timed): the first use of every entry point allocates its workspace / pinned
    # staging buffers and loads its code object; with it the timed region measures steady-state steps even for --warmup 0
    one_step()
    torch.cuda.synchronize()
    for _ in range(args.warmup):
        one_step()
    torch.cuda.synchronize()
    barrier()
    learner.events = []  # per-launch HIP events around the dominant kernels (same stream as the launches)
    evts = []
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        one_step(evts)
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device=dev)
        torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MAX, group=pg)
        dt = float(tt.item())

    phases = [0.0, 0.0, 0.0]
    for e in evts:
        for i in range(3):
            phases[i] += e[i].elapsed_time(e[i + 1])
    phases = [p / max(1, len(evts)) for p in phases]
    act_ms = [s.elapsed_time(e) for (k, s, e) in learner.events if k == "actor"]
    cri_ms = [s.elapsed_time(e) for (k, s, e) in learner.events if k == "critic"]
    learner.events = None

    if rank == 0:
        units = world * E * A * T * args.steps
        rows_a = E * A * T
        if actor_kind == "gru":  # fc1 + 6 gate blocks + head
            Pa = aspec.din * 64 + 6 * 64 * 64 + 64 * roll.K
        else:
            Pa = aspec.din * 64 + 64 * 64 + 64 * roll.K
        flop_actor = rows_a * (2 * Pa + 2 * Pa + 2 * (Pa - aspec.din * 64))  # fwd + dW + dX
        avg_actor_ms = sum(act_ms) / max(1, len(act_ms))
        achieved = flop_actor / (avg_actor_ms * 1e-3) / 1e12 if avg_actor_ms > 0 else 0.0
        out = {
            "metric": "env-steps/sec (agents x envs x steps), MAPPO full iteration", "value": units / dt,
            "unit": "agent-env-steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32" if N.load().cm_mfma_mode() == 0 else "bf16x3 (opt-in CM_MFMA=bf16x3: error-compensated bf16 MFMA products, fp32 accumulate and storage)",
            "data": "synthetic",
            "config": {"workload": desc if not args.envs else f"{desc} [envs/GPU overridden to {E}]",
                       "envs_per_gpu": E, "agents": A, "steps": T, "epochs": hp.epochs, "parallelism": f"env-sharded x{world}"},
            "ppo_update_ms": phases[2], "ppo_update_ms_per_epoch": phases[2] / hp.epochs,
            "phase_ms": {"rollout": phases[0], "value_pass_scan": phases[1], "update": phases[2]},
            "kernel_ms": {"actor_fwd_bwd": avg_actor_ms, "critic_fwd_bwd": sum(cri_ms) / max(1, len(cri_ms))},
            "roofline": {"kernel": "k_mlp<NCH,M_ACTOR> (cm_ppo_actor_fwd_bwd)" if actor_kind == "mlp" else
                         "k_gru_chunk_fwd + k_gru_chunk_bwd (all TBPTT chunks of one epoch)", "bound": "mfma", "achieved": achieved,
                         "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": achieved / PEAK_F32_MFMA_TFLOPS,
                         "traffic": None, "flop_per_launch": flop_actor},
        }
        # HBM bytes of the dominant kernel come from a separate rocprofv3 --pmc pass (counters cannot be read
        # live here); tools/pmc_summary.py writes them to profiles/pmc_dominant_kernel.json
        pmc_path = os.path.join(ROOT, "profiles", "pmc_dominant_kernel.json")
        if os.path.exists(pmc_path) and args.workload == "cfg3" and not args.envs:
            pmc = json.load(open(pmc_path))
            out["roofline"]["traffic"] = pmc["hbm_bytes_per_launch"]
            out["roofline"]["traffic_source"] = pmc["source"]
            out["roofline"]["mfma_busy_frac_pmc"] = pmc.get("mfma_busy_frac")
            out["roofline"]["algorithmic_bytes_per_launch"] = rows_a * (4 * aspec.din + roll.K + 12)
        if world == 1 and not args.no_cpu_baseline:
            from oracle import reference_loop  # checker / baseline only -- never part of the measured path
            Ec = args.cpu_envs
            r = reference_loop.run(E=Ec, A=A, T=T)
            out["cpu_baseline"] = {"value": r["agent_steps_per_s"], "unit": "agent-env-steps/s",
                                   "cores": min(os.cpu_count() or 1, Ec + r["threads"]), "kind": "port",
                                   "sample": f"one iteration of the reference-structured driver (oracle/reference_loop.py: "
                                             f"process-per-env pipes, per-step python loops) at {Ec} envs x {A} agents x {T} steps, "
                                             f"torch threads={r['threads']}; rollout {r['rollout_s']:.2f}s gae {r['gae_s']:.2f}s "
                                             f"update {r['update_s']:.2f}s"}
        print(json.dumps(out), flush=True)
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
