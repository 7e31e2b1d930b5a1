#!/usr/bin/env python3
"""bench.py -- MAPPO hot path on MI355X: rollout buffer -> TD(lambda) scan -> PPO update.

A "step" is ONE full training iteration of the reference's outer loop (cleanmarl/mappo_multienvs.py:379-612)
on the synthetic fixed-shape MPE-like env: rollout of E envs x A agents x T steps into the device-resident
buffer, the value pass + TD(lambda) scan, and `epochs` full-batch PPO updates (actor + critic fwd/bwd,
grad all-reduce when N > 1, fused norm/clip/Adam).  Inputs never leave HBM.

    python bench.py --gpus 1 --steps 5 --warmup 2
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Prints ONE JSON line on rank 0 (contract in the task statement): `value` = agents*envs*steps per second summed over all
ranks.  Scaling is STRONG by default, as BASELINE.json's north_star defines it ("num_envs shards across the 8 GPUs"): the
workload's env count is global, rank r owns envs dist.shard(E, r, N), global env index = first + e.  `--scaling weak` keeps E
envs on every rank instead.  With N > 1 the default run times the two all-reduce messages of an optimiser step alone ("allreduce_us"
in the line), prints the line, and only THEN runs a short leg in the other scaling mode, reported on stderr as
`[bench extra leg] {...}` -- an extra leg can never cost the headline record (`value` is always the mode named in "scaling").
Every N = 1 extra leg is guarded the same way ("extras_error" / "cpu_baseline_error" instead of a lost line).

Extra keys (N = 1): "phase_roofline" (rollout / value pass / critic against their own bounds), "other_workloads" (the other
BASELINE.json configs, a few steps each after the timed region) and "strong_scaling_shares" (the per-GPU share of each sharded
config timed on this one GPU: t(E) / t(E/8) bounds the 8-GPU speed-up before the all-reduces).
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
    # name: (GLOBAL envs, A, T, algo, env, actor, description) -- BASELINE.json configs[2] is the headline (the metric is
    # quoted on it) and the default; the others are parity-test cases, timed after the headline (other_workloads) or on request
    "cfg3": (4096, 8, 128, "mappo", "spread", "mlp", "MAPPO synthetic-MPE 4096 envs x 8 agents x 128 steps, 2x64 MLP (BASELINE.json configs[2])"),
    "cfg2": (1024, 3, 128, "mappo", "spread", "mlp", "MAPPO synthetic-MPE 1024 envs x 3 agents x 128 steps, 2x64 MLP (BASELINE.json configs[1])"),
    "cfg4": (2048, 10, 256, "ippo", "shape", "mlp", "IPPO smaclite-shape synthetic 2048 envs x 10 agents x 256 steps, 2x64 MLP (BASELINE.json configs[3])"),
    "cfg5": (1024, 5, 128, "mappo", "spread", "gru", "MAPPO-GRU synthetic-MPE 1024 envs x 5 agents x 128 steps, GRU hidden 64, tbptt 10 (BASELINE.json configs[4])"),
}
PEAK_F32_MFMA_TFLOPS = 157.3  # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, 256 CU x 4 SIMD x 64 FLOP/clk x 2.4 GHz
PEAK_HBM_GBS = 8000.0         # MI355X_MICROARCH.md: HBM3E spec rate


class Workload:
    """One BASELINE config on this rank's env shard: device rollout + learner, reference default hyper-parameters."""

    def __init__(self, name, E, env_offset, dev, pg=None, world=1):
        from cleanmarl_amd.gru import GRUPPOLearner, GRUSyntheticRollout
        from cleanmarl_amd.learner import HParams, NetSpec, PPOLearner, init_params_like_torch
        from cleanmarl_amd.rollout import SyntheticShapeRollout, SyntheticSpreadRollout
        _, A, T, algo, env_kind, actor_kind, desc = WORKLOADS[name]
        self.name, self.E, self.A, self.T, self.algo, self.actor_kind, self.desc = name, E, A, T, algo, actor_kind, desc
        self.world, self.pg = world, pg
        self.hp = hp = HParams()  # reference defaults: gamma .99, lambda .95, eps .2, c_ent 1e-3, lr 8e-4, epochs 3 (tbptt 10)
        if env_kind == "shape":
            self.roll = SyntheticShapeRollout(E, A, T, seed=1, agent_ids=True, device=dev, env_offset=env_offset)
        elif actor_kind == "gru":
            self.roll = GRUSyntheticRollout(E, A, T, seed=1, agent_ids=True, device=dev, env_offset=env_offset)
        else:
            self.roll = SyntheticSpreadRollout(E, A, T, seed=1, agent_ids=True, device=dev, env_offset=env_offset)
        roll = self.roll
        self.aspec = NetSpec(roll.Do, 64, 0 if actor_kind == "gru" else 1, roll.K, actor_kind)
        self.cspec = NetSpec(roll.Ds if algo == "mappo" else roll.Do, 64, 1, 1)
        torch.manual_seed(1)  # reference construction order actor -> critic (:329-339); identical on every rank
        a_init = init_params_like_torch(self.aspec)
        c_init = init_params_like_torch(self.cspec)
        self.learner = (GRUPPOLearner if actor_kind == "gru" else PPOLearner)(algo, self.aspec, self.cspec, A, hp, dev, a_init, c_init,
                                                                          pg, world)

    @staticmethod
    def message_floats(name):
        """Floats of the actor's [gradient | statistics] message of a workload (spread env: Do = 6 A + A, K = 5; 2 x 64 MLP)."""
        _, A, T, algo, env_kind, actor_kind, _ = WORKLOADS[name]
        Do, H, K = 7 * A, 64, 5
        return Do * H + H + H * H + H + H * K + K + 8

    def one_step(self, evts=None):
        L = self.learner
        if evts is not None:
            e = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
            e[0].record()
        b = self.roll.collect(L.actor, self.aspec)
        if evts is not None:
            e[1].record()
        L.compute_targets(b)
        if evts is not None:
            e[2].record()
        L.update(b)
        if evts is not None:
            e[3].record()
            evts.append(e + [getattr(L, "critic_span", None)])

    def barrier(self):
        if self.world > 1:
            torch.distributed.barrier(group=self.pg)

    def run(self, steps, warmup):
        """1 untimed setup pass + `warmup` untimed steps, then EXACTLY `steps` timed steps between barrier + synchronize pairs;
        returns wall seconds (max over ranks), per-phase event times and the per-launch times of the dominant kernels."""
        L = self.learner
        # setup pass (not one of the W warmup steps, never timed): the first use of every entry point allocates its workspace /
        # pinned staging buffers and loads its code object; with it the timed region measures steady-state steps even for W = 0
        self.one_step()
        torch.cuda.synchronize()
        for _ in range(warmup):
            self.one_step()
        torch.cuda.synchronize()
        self.barrier()
        # HIP events (per phase, and per launch around the dominant kernels on the stream they are launched on) are recorded on every
        # `stride`-th timed step only: an event record is a barrier packet between two dependent launches (~ 10 us each in the kernel
        # trace of the 512-env share: 16 records = 2 - 3 % of a 1.4 ms iteration; the CLI, which records none, showed no such gaps,
        # profiles/r03_timeline_cli_envs512.txt).  >= 5 instrumented steps whenever K >= 5; all of them for K < 10.
        stride = max(1, min(4, steps // 5))
        kernel_events, evts = [], []
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            sampled = i % stride == 0
            L.events = kernel_events if sampled else None
            self.one_step(evts if sampled else None)
        L.events = kernel_events
        torch.cuda.synchronize()
        self.barrier()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        if self.world > 1:
            tt = torch.tensor([dt], dtype=torch.float64, device=L.device)
            torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MAX, group=self.pg)
            dt = float(tt.item())
        # update = the longer of the two halves: actor epochs on the launch stream (e2 -> e3), critic epochs on their own stream
        # (e2 -> end of the last critic step; they overlap the NEXT step's rollout, learner.PPOLearner.update)
        phases = [0.0, 0.0, 0.0, 0.0]
        for e in evts:
            for i in range(3):
                phases[i] += e[i].elapsed_time(e[i + 1])
            if e[4] is not None:
                phases[3] += e[2].elapsed_time(e[4][1])
        phases = [p / max(1, len(evts)) for p in phases]
        phases[2], actor_half = max(phases[2], phases[3]), phases[2]
        act_ms = [s.elapsed_time(e) for (k, s, e) in L.events if k == "actor"]
        cri_ms = [s.elapsed_time(e) for (k, s, e) in L.events if k == "critic"]
        L.events = None
        mean = lambda v: sum(v) / max(1, len(v))
        return dict(dt=dt, steps=steps, ms_per_step=1e3 * dt / steps, phase_ms=dict(rollout=phases[0], value_pass_scan=phases[1],
                    update=phases[2], update_actor_stream=actor_half, update_critic_stream=phases[3]),
                    actor_ms=mean(act_ms), critic_ms=mean(cri_ms))

    def solo_actor_leg(self, launches=10, warm=3):
        """The dominant kernel ALONE on the device: `launches` back-to-back cm_ppo_actor_fwd_bwd_ld calls (the pass + its fold launch; no
        optimiser step, the parameters do not move) on a fresh batch with nothing else in flight, each bracketed by HIP events on the launch
        stream.  The timed iterations run the critic's epochs beside the actor's (learner.overlap_critic: the faster schedule), so the
        in-iteration duration of this kernel is a co-residency figure; the roofline of the kernel itself is quoted on this leg.
        -> (mean ms, [ms per launch]) or None for workloads without an MLP actor pass."""
        from cleanmarl_amd import _native as N
        L = self.learner
        if self.actor_kind != "mlp" or self.E == 0:
            return None
        # the rollout and the value pass + scan alone as well (phase_roofline: inside the timed iterations the critic's last epoch is still
        # running beside the next rollout, so their in-iteration event times are co-residency figures too)
        torch.cuda.synchronize()
        ro, vp = [], []
        for _ in range(3):
            e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
            e0.record(); b = self.roll.collect(L.actor, self.aspec); e1.record(); L.compute_targets(b); e2.record()
            torch.cuda.synchronize()
            ro.append(e0.elapsed_time(e1)); vp.append(e1.elapsed_time(e2))
        self.solo_rollout_ms, self.solo_value_pass_ms = min(ro), min(vp)
        s = N.stream_ptr()
        for _ in range(warm):
            L.actor_pass(b, s)
        torch.cuda.synchronize()
        ev = []
        for _ in range(launches):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); L.actor_pass(b, s); e1.record()
            ev.append((e0, e1))
        torch.cuda.synchronize()
        ms = [a.elapsed_time(c) for a, c in ev]
        # the critic's pass the same way (phase_roofline.critic_fwd_bwd: beside the actor's kernels it only fills their tails): the plain pass of
        # epochs 2 .. n, and -- where the value pass hands its layer-0 activations over (learner._keeps_h0) -- the first epoch's pass without the W0 product
        L.wait_critic()
        h0_key = getattr(L, "_h0_key", None)

        def critic_alone():
            for _ in range(warm):
                L.critic_pass(b, s)
            torch.cuda.synchronize()
            evc = []
            for _ in range(launches):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); L.critic_pass(b, s); e1.record()
                evc.append((e0, e1))
            torch.cuda.synchronize()
            return sum(a.elapsed_time(c) for a, c in evc) / len(evc)

        self.solo_critic_first_ms = critic_alone() if h0_key is not None else None
        L._h0_key = None
        self.solo_critic_ms = critic_alone()
        L._h0_key = h0_key
        return sum(ms) / len(ms), ms

    # ---- algorithmic work per launch (weights, MFMA-tile padding and recomputation are NOT counted)
    def work(self):
        E, A, T, K = self.E, self.A, self.T, self.roll.K
        Do, Dc, H = self.aspec.din, self.cspec.din, 64
        rows_a = E * A * T
        rows_c = E * T * (1 if self.algo == "mappo" else A)
        if self.actor_kind == "gru":  # fc1 + 6 gate blocks + head
            Pa = Do * H + 6 * H * H + H * K
        else:
            Pa = Do * H + H * H + H * K
        Pc = Dc * H + H * H + H
        return dict(
            rows_a=rows_a, rows_c=rows_c,
            actor=dict(flop=rows_a * (2 * Pa + 2 * Pa + 2 * (Pa - Do * H)), bytes=rows_a * (4 * Do + K + 12)),
            critic=dict(flop=rows_c * (4 * Pc + 2 * (Pc - Dc * H)), bytes=rows_c * (4 * Dc + 4)),
            value_pass=dict(flop=rows_c * 2 * Pc, bytes=rows_c * (4 * Dc + 4) + E * T * (8 + 8 * A)),  # + the scan's reads / writes
            rollout=dict(flop=rows_a * 2 * Pa, bytes=E * T * (4 * A * Do + 4 * self.roll.Ds + 4 + 8 * A)))

    def close(self):
        self.roll = self.learner = None
        torch.cuda.empty_cache()


def one_rank_rccl_floor_us(dev, n_floats, reps=200):
    """The actor's [gradient | statistics] message as an all-reduce on a ONE-rank RCCL communicator, microseconds on the stream: the
    latency floor of the exchange step (launch + protocol, no link).  -> (us or None, description of the source, measured: bool)."""
    import socket
    created = False
    try:
        if not torch.distributed.is_initialized():
            with socket.socket() as sk:
                sk.bind(("127.0.0.1", 0))
                port = sk.getsockname()[1]
            torch.distributed.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=dev)
            created = True
        buf = torch.zeros(n_floats, dtype=torch.float32, device=dev)
        for _ in range(30):
            torch.distributed.all_reduce(buf)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            torch.distributed.all_reduce(buf)
        e1.record()
        torch.cuda.synchronize()
        us = 1e3 * e0.elapsed_time(e1) / reps
        return us, f"one-rank RCCL all-reduce of {4 * n_floats} B timed on this GPU ({reps} back-to-back calls, HIP events): a floor, no xGMI hop", True
    except Exception as ex:  # noqa: BLE001 -- no live number: the caller reports the projection without communication only
        return None, f"live one-rank RCCL measurement failed: {ex!r}"[:300], False
    finally:
        if created:  # only the group this function made: never tear down a caller's
            try:
                torch.distributed.destroy_process_group()
            except Exception:  # noqa: BLE001
                pass


DOMINANT_KERNEL = "k_mlp<1, 2,"  # cm_ppo_actor_fwd_bwd at config 3: one input chunk, M_ACTOR (the name rocprofv3 prints starts "... k_mlp<1, 2, ...")


def load_pmc(path, workload="cfg3"):
    """roofline.traffic comes from rocprofv3 --pmc passes that cannot run inside this process (profiles/pmc_dominant_kernel.json, written
    by tools/pmc_summary.py --emit).  It is only valid for the kernels it was measured on: the record carries the hash of the library's
    sources (cleanmarl_amd/build.py::source_hash) and the profiled kernel's name; a record taken on other sources, of another kernel or
    for another workload is refused.  -> (record or None, reason it was refused or None)."""
    from cleanmarl_amd.build import source_hash
    if not os.path.exists(path):
        return None, "no PMC record (profiles/pmc_dominant_kernel.json missing)"
    try:
        pmc = json.load(open(path))
    except Exception as ex:  # noqa: BLE001
        return None, f"unreadable PMC record: {ex!r}"[:200]
    if pmc.get("workload") != workload:
        return None, f"PMC record is for workload {pmc.get('workload')!r}, not {workload!r}"
    if DOMINANT_KERNEL not in str(pmc.get("kernel", "")):
        return None, f"PMC record is for kernel {str(pmc.get('kernel'))[:80]!r}, not the actor pass {DOMINANT_KERNEL!r}"
    have, want = pmc.get("source_hash"), source_hash()
    if have != want:
        return None, f"stale PMC record: taken on sources {have}, the library being timed is built from {want} (re-run tools/refresh_profiles.sh)"
    return pmc, None


def load_issue_counters(path, kernel_pat):
    """Instruction-issue counters of the dominant kernel (profiles/issue_counters.json, tools/gpu/r06_issue_pmc.sh: rocprofv3 --pmc passes),
    accepted only for the sources of the library being timed.  On gfx950 fp32 MFMAs and VALU instructions do not co-execute (DESIGN.md
    section 3, tools/probes/cosimd_overlap.hip), so a kernel's issue bound is MFMA cycles + VALU instruction cycles per SIMD; this turns the
    counters into those shares of the launch.  -> dict or None."""
    from cleanmarl_amd.build import source_hash
    try:
        rec = json.load(open(path))
    except Exception:  # noqa: BLE001
        return None
    if rec.get("source_hash") != source_hash():
        return None
    k = next((v for n, v in rec.get("kernels", {}).items() if kernel_pat in n), None)
    if not k or not k.get("GRBM_GUI_ACTIVE"):
        return None
    simd_cycles = 1024.0 * k["GRBM_GUI_ACTIVE"] / 8.0          # 256 CUs x 4 SIMDs x the launch's cycles (GRBM_GUI_ACTIVE is summed over 8 XCDs)
    valu = k["SQ_INSTS_VALU"] - k["SQ_INSTS_MFMA"]             # SQ_INSTS_VALU counts the MFMAs as well
    return {"mfma_insts_per_launch": k["SQ_INSTS_MFMA"], "other_valu_insts_per_launch": valu,
            "mfma_busy_frac": k["SQ_VALU_MFMA_BUSY_CYCLES"] / simd_cycles,
            "valu_issue_frac_at_4_cycles": 4.0 * valu / simd_cycles, "valu_issue_frac_at_7_cycles": 7.0 * valu / simd_cycles,
            "valu_mfma_coexec_cycles": k.get("SQ_VALU_MFMA_COEXEC_CYCLES"),
            "note": "shares of the launch's SIMD cycles: MFMA pipe busy, and the other VALU instructions at 4 cycles each (issue minimum of a 64-lane "
                    "wave) / 7 cycles each (marginal cost measured behind an MFMA, profiles/r06_cosimd_overlap.txt); the two do not co-execute, so their "
                    "sum is the part of the launch its instruction stream needs whatever the schedule",
            "source": "profiles/r06_issue_counters.txt (rocprofv3 --pmc, tools/gpu/r06_issue_pmc.sh), stamped with the library's source hash"}


def _bound(work, ms):
    """Achieved rates of one launch / phase against BOTH peaks; the bound is the one that takes longer at peak."""
    if ms <= 0:
        return None
    tf, gb = work["flop"] / (ms * 1e-3) / 1e12, work["bytes"] / (ms * 1e-3) / 1e9
    t_mfma, t_hbm = work["flop"] / (PEAK_F32_MFMA_TFLOPS * 1e12), work["bytes"] / (PEAK_HBM_GBS * 1e9)
    b = "mfma" if t_mfma >= t_hbm else "hbm"
    return dict(ms=ms, bound=b, tflops=tf, gbs=gb, frac=(tf / PEAK_F32_MFMA_TFLOPS if b == "mfma" else gb / PEAK_HBM_GBS))


def summarize(w, r, solo=None):
    """Compact record of one timed workload (used for other_workloads / strong_scaling_shares).  solo: Workload.solo_actor_leg()'s result --
    roofline_frac is then the actor pass's own rate (alone on the device), `actor_fwd_bwd_ms` stays the in-iteration duration."""
    wk = w.work()
    units = w.world * w.E * w.A * w.T * r["steps"]
    kern_ms = solo[0] if solo else r["actor_ms"]
    out = dict(envs=w.E, ms_per_step=r["ms_per_step"], value=units / r["dt"], phase_ms=r["phase_ms"],
               actor_fwd_bwd_ms=r["actor_ms"], critic_fwd_bwd_ms=r["critic_ms"],
               roofline_frac=(_bound(wk["actor"], kern_ms) or {}).get("frac"))
    if solo:
        out["actor_fwd_bwd_solo_ms"] = solo[0]
        out["critic_fwd_bwd_solo_ms"] = getattr(w, "solo_critic_ms", None)
    return out


def message_latencies(w, dev, pg, N, peer_too=True):
    """The two messages of an optimiser step timed alone, on the stream, in microseconds: as RCCL all-reduces (+ the stand-alone
    optimiser-step launch that follows one, for a like-for-like sum) and as the one-shot peer exchange fused with the step
    (dist.PeerAllReduce: push + fold / step; SGD with lr = 0, so nothing moves)."""
    from cleanmarl_amd import dist as D
    lat = {"rccl": {}, "optimizer_step": {}, "peer_exchange_plus_step": {}}
    lib = N.load()

    def timed(fn, reps=100, warm=20):
        for _ in range(warm):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return 1e3 * e0.elapsed_time(e1) / reps

    for nm, npar in (("actor", w.aspec.nparams), ("critic", w.cspec.nparams)):
        n = npar + N.NUM_STATS
        key = f"{nm}_{4 * n}B"
        buf = torch.zeros(n, dtype=torch.float32, device=dev)
        buf[npar + N.STAT_COUNT] = 1.0
        params = torch.zeros(npar, dtype=torch.float32, device=dev)
        norm = torch.zeros(1, dtype=torch.float32, device=dev)
        scratch = torch.zeros(lib.cm_opt_step_scratch_bytes(), dtype=torch.uint8, device=dev)
        step = [0]

        def opt():
            step[0] += 1
            return N.OptStep(params=params.data_ptr(), exp_avg=0, exp_avg_sq=0, out_norm=norm.data_ptr(), scratch=scratch.data_ptr(), lr=0.0,
                             beta1=0.9, beta2=0.999, eps=1e-8, weight_decay=0.0, max_norm=-1.0, grad_scale=1.0, step=step[0], opt_kind=N.OPT_SGD)
        lat["rccl"][key] = timed(lambda: torch.distributed.all_reduce(buf, group=pg))
        lat["optimizer_step"][key] = timed(lambda: N.check(lib.cm_optimizer_step(N.ptr(buf), npar, opt(), N.stream_ptr()), "cm_optimizer_step"))
        if peer_too:
            try:
                peer = D.PeerAllReduce(n, pg)
                # one exchange first, waited for, with a short wall-time bound: a mailbox mapping that does not reach its peer shows up here as
                # an error after 5 s (the step kernel gives up, check() raises) instead of 120 timed launches giving up one after the other
                peer.timeout_s = min(peer.timeout_s, 5.0)
                peer.step(buf, npar, opt(), N.stream_ptr())
                torch.cuda.synchronize()
                peer.check()
                lat["peer_exchange_plus_step"][key] = timed(lambda: peer.step(buf, npar, opt(), N.stream_ptr()))
                peer.check()
                peer.close()
            except Exception as ex:  # noqa: BLE001 -- e.g. a node without peer access: reported, not fatal
                lat["peer_exchange_plus_step"][key] = repr(ex)[:200]
    return lat


def total_envs_of(args, world, E_glob):
    return E_glob if args.scaling == "strong" else world * E_glob


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="cfg3", choices=sorted(WORKLOADS))
    ap.add_argument("--scaling", default="strong", choices=["strong", "weak"],
                    help="strong (default, north_star): the workload's envs are sharded over the ranks; weak: every rank owns all of them")
    ap.add_argument("--envs", type=int, default=0, help="override the env count (global in strong mode, per GPU in weak mode)")
    ap.add_argument("--allreduce", default="rccl", choices=["rccl", "peer"],
                    help="exchange step of the MLP learner at N > 1: RCCL all-reduce + optimiser step (default) or the one-shot peer "
                         "all-reduce over hipIpc mailboxes fused with the step (csrc/cm_peer.hip); the same flag on every rank")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip other_workloads / strong_scaling_shares / weak_scaling legs")
    ap.add_argument("--cpu-envs", type=int, default=256, help="envs in the bounded CPU-baseline sample")
    ap.add_argument("--solo-launches", type=int, default=10,
                    help="launches of the solo leg that times the dominant kernel alone after the timed region (N = 1; 0 = off)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # a bare `python bench.py --gpus N`: start the N ranks ourselves, exactly as the driver's command line does (one process per GPU
        # under torch.distributed.run); rank 0's JSON line passes through on stdout
        import socket
        import subprocess
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.call(cmd))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus}: launched with WORLD_SIZE={world}")
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
    from cleanmarl_amd import dist

    # rank-invariant by construction: a command-line flag, identical on every rank (the learner reads it at construction)
    os.environ["CM_PEER_ALLREDUCE"] = "1" if args.allreduce == "peer" else "0"
    E_glob = args.envs or WORKLOADS[args.workload][0]
    if args.scaling == "strong":
        if E_glob < world:
            raise SystemExit(f"{E_glob} envs cannot be sharded over {world} ranks")
        first, E = dist.shard(E_glob, rank, world)
    else:
        first, E = rank * E_glob, E_glob
    w = Workload(args.workload, E, first, dev, pg, world)
    w.learner.global_envs = total_envs_of(args, world, E_glob)  # rank-invariant schedule choice (learner._schedule_rows)
    # the shader clock of the dominant kernel, read by the kernel itself (cm_clock_probe: workgroup 0 of every actor pass leaves s_memtime /
    # s_memrealtime at entry and exit of its tile loop; the last launch of the timed region is what is read back)
    clk = torch.zeros(512, 4, dtype=torch.int64, device=dev)
    N.check(N.load().cm_clock_probe(N.ptr(clk)), "cm_clock_probe")
    r = w.run(args.steps, args.warmup)
    # the dominant kernel alone (N = 1, MLP actors): after the timed region, before the probe is read -- the clock / workgroup-span figures
    # below then describe a solo launch as well.  --solo-launches 0 skips the leg (the roofline then falls back to the in-iteration events)
    solo = None
    if world == 1 and args.solo_launches > 0:
        try:
            solo = w.solo_actor_leg(args.solo_launches)
        except Exception as ex:  # noqa: BLE001 -- never instead of the headline line
            solo = None
            print(f"[bench] solo leg failed: {ex!r}", file=sys.stderr)
    N.check(N.load().cm_clock_probe(None), "cm_clock_probe")
    clk_all = clk.cpu()
    clk = [int(v) for v in clk_all[0]]
    A, T, hp = w.A, w.T, w.hp
    total_envs = E_glob if args.scaling == "strong" else world * E_glob

    out = None
    if rank == 0:
        units = total_envs * A * T * args.steps
        wk = w.work()
        # the kernel's own duration: the solo leg when there is one, else the launches inside the timed iterations
        kern_ms = solo[0] if solo else r["actor_ms"]
        act = _bound(wk["actor"], kern_ms) or dict(tflops=0.0)
        achieved = act["tflops"]
        desc = w.desc if not args.envs else f"{w.desc} [envs overridden to {E_glob}]"
        out = {
            "metric": "env-steps/sec (agents x envs x steps), MAPPO full iteration", "value": units / r["dt"],
            "unit": "agent-env-steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": r["ms_per_step"], "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
            "dtype": {0: "f32", 1: "bf16x3 (opt-in CM_MFMA=bf16x3: error-compensated bf16 MFMA products, fp32 accumulate and storage)",
                      2: "bf16 (opt-in CM_MFMA=bf16: single-pass bf16 MFMA products in the training passes, fp32 accumulate and storage; looser parity tier)"}[N.load().cm_mfma_mode()],
            "data": "synthetic",
            "config": {"workload": desc, "global_envs": total_envs, "envs_per_gpu": E, "agents": A, "steps": T, "epochs": hp.epochs,
                       "parallelism": f"env-sharded x{world} ({args.scaling} scaling), one all-reduce per network and optimiser step",
                       "allreduce": args.allreduce if world > 1 else "none (one rank: the optimiser step rides on the pass's reduction launch)"},
            "ppo_update_ms": r["phase_ms"]["update"], "ppo_update_ms_per_epoch": r["phase_ms"]["update"] / hp.epochs,
            "phase_ms": r["phase_ms"],
            "phase_solo_ms": {"rollout": getattr(w, "solo_rollout_ms", None), "value_pass_scan": getattr(w, "solo_value_pass_ms", None),
                              "actor_fwd_bwd": solo[0] if solo else None, "critic_fwd_bwd": getattr(w, "solo_critic_ms", None),
                              "critic_first_epoch": getattr(w, "solo_critic_first_ms", None),
                              "note": "each phase alone on the device (solo leg after the timed region); phase_ms / kernel_ms are event times inside the "
                                      "timed iterations, where the critic's epochs run beside the actor's passes and the next rollout"},
            "kernel_ms": {"actor_fwd_bwd": r["actor_ms"], "critic_fwd_bwd": r["critic_ms"]},
            "roofline": {"kernel": "k_mlp<NCH,M_ACTOR> (cm_ppo_actor_fwd_bwd)" if w.actor_kind == "mlp" else
                         "k_gru_chunk_fwd + k_gru_chunk_bwd (all TBPTT chunks of one epoch)", "bound": "mfma", "achieved": achieved,
                         "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": achieved / PEAK_F32_MFMA_TFLOPS,
                         "traffic": None, "flop_per_launch": wk["actor"]["flop"],
                         "launch_ms": kern_ms, "in_iteration_launch_ms": r["actor_ms"],
                         "source": (f"solo leg: {len(solo[1])} back-to-back launches of the pass alone on the device after the timed region, HIP events on the launch "
                                    "stream (min %.4f max %.4f ms); inside the timed iterations the critic's epochs run beside it (in_iteration_launch_ms)"
                                    % (min(solo[1]), max(solo[1]))) if solo else "HIP events around the launches inside the timed iterations"},
            # the other phases of the step against their own bounds (algorithmic work of SURVEY.md §8(d) / event time)
            "phase_roofline": {"rollout": _bound(wk["rollout"], getattr(w, "solo_rollout_ms", None) or r["phase_ms"]["rollout"]),
                               "value_pass_scan": _bound(wk["value_pass"], getattr(w, "solo_value_pass_ms", None) or r["phase_ms"]["value_pass_scan"]),
                               "critic_fwd_bwd": _bound(wk["critic"], getattr(w, "solo_critic_ms", None) or r["critic_ms"]),
                               "whole_step": _bound(dict(flop=wk["rollout"]["flop"] + wk["value_pass"]["flop"] + hp.epochs * (wk["actor"]["flop"] + wk["critic"]["flop"]),
                                                         bytes=wk["rollout"]["bytes"] + wk["value_pass"]["bytes"] + hp.epochs * (wk["actor"]["bytes"] + wk["critic"]["bytes"])),
                                                    r["ms_per_step"])},
        }
        # HBM bytes of the dominant kernel come from a separate rocprofv3 --pmc pass (counters cannot be read
        # live here); tools/pmc_summary.py writes them to profiles/pmc_dominant_kernel.json
        out["roofline"]["algorithmic_bytes_per_launch"] = wk["actor"]["bytes"]
        if w.actor_kind == "mlp":
            # what the kernel ISSUES on the matrix pipe (tile padding included: Do -> 64-column chunks in dW0, K -> 16 head rows ...) beside the
            # algorithmic flop the fraction above is quoted on: issued / algorithmic is the padding, issued_frac the pipe's share of nominal peak
            per_row = N.load().cm_ppo_actor_issued_flop_per_row(w.aspec.din, w.aspec.hidden, w.aspec.n_layers, w.aspec.dout)
            if per_row > 0 and kern_ms > 0:
                issued = per_row * wk["rows_a"]
                out["roofline"]["issued_flop_per_launch"] = issued
                out["roofline"]["issued_over_algorithmic"] = issued / wk["actor"]["flop"]
                out["roofline"]["issued_frac"] = issued / (kern_ms * 1e-3) / 1e12 / PEAK_F32_MFMA_TFLOPS
        if clk[3] > clk[2] and clk[0] > 0:
            # s_memtime ticks are shader cycles, s_memrealtime ticks the constant 100 MHz reference: the clock workgroup 0 of the LAST actor
            # pass of the timed region ran at, and the pass's rate against the fp32 MFMA peak AT THAT CLOCK (256 CUs x 256 flop per cycle) --
            # `frac` above stays quoted on the nominal 2.4 GHz peak of MI355X_MICROARCH.md
            ghz = clk[0] / ((clk[3] - clk[2]) / 1e8) / 1e9
            out["roofline"]["shader_clock_ghz"] = ghz
            out["roofline"]["probe_workgroup_ms"] = (clk[3] - clk[2]) / 1e5
            out["roofline"]["frac_of_peak_at_shader_clock"] = achieved / (256 * 256 * ghz * 1e-3)
            # ramp and tail of the launch from every workgroup's reference-clock stamps (ms relative to the first workgroup's entry)
            live = clk_all[clk_all[:, 3] > 0].double()
            t0 = float(live[:, 2].min())
            st, en = (live[:, 2] - t0) / 1e5, (live[:, 3] - t0) / 1e5
            out["roofline"]["workgroup_span"] = {"workgroups": int(live.shape[0]), "last_entry_ms": float(st.max()), "first_exit_ms": float(en.min()),
                                                 "median_exit_ms": float(en.median()), "last_exit_ms": float(en.max()),
                                                 "mean_busy_ms": float((en - st).mean()), "launch_ms_hip_events": kern_ms}
        if args.workload == "cfg3" and not args.envs and world == 1:
            pmc, why = load_pmc(os.path.join(ROOT, "profiles", "pmc_dominant_kernel.json"))
            if pmc is not None:
                out["roofline"]["traffic"] = pmc["hbm_bytes_per_launch"]
                out["roofline"]["traffic_source"] = pmc["source"]
                out["roofline"]["mfma_busy_frac_pmc"] = pmc.get("mfma_busy_frac")
            else:
                out["roofline"]["traffic_refused"] = why  # traffic stays null: never a number measured on other kernels
            issue = load_issue_counters(os.path.join(ROOT, "profiles", "issue_counters.json"), DOMINANT_KERNEL)
            if issue is not None:
                out["roofline"]["issue"] = issue
    w.close()

    weak_leg = None
    if not args.no_extras and world > 1:
        # the two messages of an optimiser step, timed alone (latency-bound: 33 KB / 116 KB at cfg 3).  Plain all-reduces on the group the
        # timed region just used; rank 0's line is printed right after them and BEFORE the leg in the other scaling mode, so that nothing
        # that leg does on a node this code has not seen yet (new buffers, 8 x the envs) can cost the headline record
        lat = {}
        try:
            lat = message_latencies(w, dev, pg, N, peer_too=False)
        except Exception as ex:  # noqa: BLE001 -- reported in the line, never instead of it
            lat = {"error": repr(ex)[:300]}
        if rank == 0:
            out["allreduce_us"] = lat
        # The peer exchange has never crossed xGMI before the first multi-GPU run of this file.  It is measured under a watchdog: if it
        # has not finished within 90 s, rank 0 prints the headline line as it stands (marked) and every rank exits -- a stuck exchange
        # can cost its own numbers only.  (The step kernel's wait for the peers' tags is itself bounded: csrc/cm_optim.hip.)
        import threading

        def bail():
            if rank == 0:
                out["allreduce_us"]["peer_exchange_plus_step"] = "watchdog: not finished within 90 s"
                print(json.dumps(out), flush=True)
            os._exit(0)
        dog = threading.Timer(90.0, bail)
        dog.daemon = True
        dog.start()
        try:
            lat["peer_exchange_plus_step"] = message_latencies(w, dev, pg, N, peer_too=True)["peer_exchange_plus_step"]
        except Exception as ex:  # noqa: BLE001
            lat["peer_exchange_plus_step"] = repr(ex)[:300]
        dog.cancel()
        if rank == 0:
            # what the exchange costs INSIDE the iteration: time on the launch stream per epoch that is not the actor's pass (its
            # all-reduce / peer exchange + the optimiser step + launch gaps; the critic's messages travel on the critic's stream)
            ep = max(1, hp.epochs)
            out["comm_exposure_us"] = {"actor_stream_per_epoch_outside_the_pass": 1e3 * (r["phase_ms"]["update_actor_stream"] / ep - r["actor_ms"]),
                                       "note": "update_actor_stream / epochs - actor_fwd_bwd (HIP events of the instrumented steps)"}
            print(json.dumps(out), flush=True)
        other = "weak" if args.scaling == "strong" else "strong"
        if other == "weak":
            w2 = Workload(args.workload, E_glob, rank * E_glob, dev, pg, world)
            tot2 = world * E_glob
        else:
            f2, e2 = dist.shard(E_glob, rank, world)
            w2 = Workload(args.workload, e2, f2, dev, pg, world)
            tot2 = E_glob
        w2.learner.global_envs = tot2  # rank-invariant schedule choice, as for the timed workload
        r2 = w2.run(max(3, min(args.steps, 10)), 2)
        w2.close()
        if rank == 0:
            weak_leg = {"leg": f"{other}_scaling", "n_gpus": world, "value": tot2 * A * T * r2["steps"] / r2["dt"],
                        "ms_per_step": r2["ms_per_step"], "global_envs": tot2, "steps": r2["steps"], "phase_ms": r2["phase_ms"]}
            # NOT a second line on stdout (the contract is ONE JSON line there): stderr, tagged
            print("[bench extra leg] " + json.dumps(weak_leg), file=sys.stderr, flush=True)

    if rank == 0 and world == 1 and not args.no_extras and not args.envs:
        try:
            # the other BASELINE configs (parity-test cases) and the per-GPU shares of the sharded ones, 20 steps each, after the
            # timed region of the headline config
            others, shares = {}, {}
            full_ms = {args.workload: out["ms_per_step"]}
            for name in ("cfg2", "cfg3", "cfg4", "cfg5"):
                if name == args.workload:
                    continue
                ww = Workload(name, WORKLOADS[name][0], 0, dev)
                rr = ww.run(20, 5)  # 20 steps: a 5-step leg charges the deferred critic epochs' tail (joined at the final sync) to too few steps
                others[name] = dict(summarize(ww, rr, ww.solo_actor_leg(args.solo_launches) if args.solo_launches > 0 else None), workload=ww.desc)
                full_ms[name] = rr["ms_per_step"]
                ww.close()
            for name in ("cfg3", "cfg4"):  # the configs north_star shards over 8 GPUs
                Eg = WORKLOADS[name][0]
                rec = {}
                for g in (2, 4, 8):
                    ww = Workload(name, Eg // g, 0, dev)
                    rr = ww.run(20, 6)
                    rec[f"1/{g} ({Eg // g} envs)"] = dict(ms_per_step=rr["ms_per_step"], phase_ms=rr["phase_ms"],
                                                          speedup_bound=full_ms[name] / rr["ms_per_step"])
                    ww.close()
                shares[name] = dict(full_ms_per_step=full_ms[name], shares=rec,
                                    note="one GPU's share timed on ONE GPU: full / share bounds the N-GPU speed-up before the all-reduces")
            out["other_workloads"] = others
            out["strong_scaling_shares"] = shares
            # what the shares project for the 8-GPU node north_star shards over, communication INCLUDED: the actor's message of every epoch
            # is exposed (the next actor pass needs the step), the critic's travel on their own stream.  L = the 33 KB actor message as a
            # one-rank RCCL all-reduce timed here, on the stream -- a FLOOR (no xGMI hop: a real 8-rank all-reduce cannot be faster); at
            # N > 1 the line carries the measured "allreduce_us" instead and the driver computes the real speed-up from its own clock
            if "cfg3" in shares:
                L_us, src, measured = one_rank_rccl_floor_us(dev, Workload.message_floats("cfg3"))
                sh8 = [v for k, v in shares["cfg3"]["shares"].items() if k.startswith("1/8")][0]["ms_per_step"]
                full = shares["cfg3"]["full_ms_per_step"]
                proj = {"without_communication": full / sh8, "full_ms": full, "share_ms": sh8, "exposed_messages_per_iteration": hp.epochs,
                        "latency_measured": measured, "latency_source": src,
                        "formula": "full / (share + epochs x latency): the three actor messages are exposed, the critic's ride on the critic stream"}
                if measured:  # no constant stands in for a failed measurement: then only the communication-free bound is reported
                    proj["value"] = full / (sh8 + hp.epochs * L_us * 1e-3)
                    proj["latency_us"] = L_us
                    # the one-rank number is a floor; the same projection at a realistic 8-rank xGMI latency, stated beside it
                    proj["at_30us"] = full / (sh8 + hp.epochs * 30e-3)
                out["projected_speedup_8"] = proj

        except Exception as ex:  # noqa: BLE001 -- an extra leg must never cost the headline line
            out["extras_error"] = repr(ex)[:500]
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        try:
            from oracle import reference_loop  # checker / baseline only -- never part of the measured path
            Ec, cores = args.cpu_envs, os.cpu_count() or 1
            sweep = sorted({1, min(8, cores), min(32, cores)})  # never all cores: 256 intra-op threads took 470 s for ONE probe epoch
            rc = reference_loop.run(E=Ec, A=A, T=T, thread_sweep=sweep)
            th = rc["threads"]
            out["cpu_baseline"] = {"value": rc["agent_steps_per_s"], "unit": "agent-env-steps/s",
                                   "cores": min(cores, Ec + max(th.values())), "kind": "port",
                                   "sample": f"one iteration of the reference-structured driver (oracle/reference_loop.py: "
                                             f"process-per-env pipes, per-step python loops) at {Ec} envs x {A} agents x {T} steps on a "
                                             f"{cores}-core host; torch intra-op threads swept over {sweep} per phase on a slice of the batch "
                                             f"and the fastest kept (rollout {th['rollout']}, scan {th['gae']}, update {th['update']}; torch's "
                                             f"default = all cores is the slowest for these tiny ops); rollout {rc['rollout_s']:.2f}s gae "
                                             f"{rc['gae_s']:.2f}s update {rc['update_s']:.2f}s",
                                   "thread_probe_s": rc["thread_probes"]}
        except Exception as ex:  # noqa: BLE001 -- an extra leg must never cost the headline line
            out["cpu_baseline_error"] = repr(ex)[:500]
    if rank == 0 and (world == 1 or args.no_extras):
        print(json.dumps(out), flush=True)  # N > 1 with extras: already printed above, before the extra leg
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
