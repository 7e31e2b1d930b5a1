"""N = 2 product path on ONE MI355X: two processes share cuda:0, each owns half of the envs of a golden batch,
runs the real PPOLearner (HIP kernels through the C-ABI) and the real collectives of cleanmarl_amd/dist.py --
over gloo here, because RCCL refuses two ranks on one device; the learner code is backend-agnostic -- and must
land on what the UNMODIFIED single-process reference produced for the full batch (tests/golden/*.npz):
post-update parameters of every epoch, returns/advantages of the shard, logged scalars.  This is the
"env sharding == full batch" claim of SURVEY.md 8(e) checked end to end on the hardware."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from parity import DISP_TOL, GRAD_TOL, TOL, StepChecker, check_grads, check_step, disp_err, golden_before, golden_init, grad_err  # noqa: F401
from parity import err as _err

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _check_final_params(got, z, last=True):
    """got["actor"] / got["critic"] after the update against the golden's parameters after its last (or only) optimiser step, as the
    displacement of that step on the well-conditioned entries (tests/parity.py; per-step gradients do not travel back from the ranks here --
    test_env_shards_reproduce_the_single_process_reference holds every step of a sharded run to the optimiser twin as well)."""
    for net in ("actor", "critic"):
        k = len(z[net + "_after"]) - 1 if last else 0
        check_step(got[net], z[net + "_after"][k], golden_before(z, net, k), "dist final " + net + " step", ref_grad=z[net + "_grads"][k])


def _worker(rank, world, port, gold, algo, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.distributed.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import restatement as R
    from cleanmarl_amd import dist
    from cleanmarl_amd.learner import DeviceBatch, HParams, NetSpec, PPOLearner
    batch, ap, cp, hp, z = R.load_golden(gold)
    dev = torch.device("cuda:0")
    reward = torch.from_numpy(z["b_reward_raw"]) if "b_reward_raw" in z.files else batch["reward"]
    lo, n = dist.shard(batch["obs"].shape[0], rank, world)
    sl = slice(lo, lo + n)
    b = DeviceBatch.from_reference_layout(batch["obs"][sl], batch["actions"][sl], batch["log_probs"][sl], reward[sl],
                                          batch["states"][sl], batch["avail"][sl], batch["mask"][sl], dev)
    H = HParams(gamma=hp["gamma"], td_lambda=hp["td_lambda"], normalize_reward=bool(hp["normalize_reward"]),
                normalize_advantage=bool(hp["normalize_advantage"]), normalize_return=bool(hp["normalize_return"]),
                epochs=int(hp["epochs"]), ppo_clip=hp["ppo_clip"], entropy_coef=hp["entropy_coef"],
                clip_gradients=hp["clip_gradients"], optimizer=hp["optimizer"],
                learning_rate_actor=hp["learning_rate_actor"], learning_rate_critic=hp["learning_rate_critic"])
    aspec = NetSpec(ap[0].shape[1], ap[0].shape[0], len(ap) // 2 - 2, ap[-1].shape[0])
    cspec = NetSpec(cp[0].shape[1], cp[0].shape[0], len(cp) // 2 - 2, 1)
    L = PPOLearner(algo, aspec, cspec, batch["obs"].shape[2], H, dev, actor_params=ap, critic_params=cp,
                   process_group=torch.distributed.group.WORLD, world_size=world)
    recs = L.train_iteration(b, keep_grads=True)
    res = dict(lo=lo, n=n, ret=b.ret.permute(0, 2, 1).cpu(), adv=b.adv.permute(0, 2, 1).cpu(),
               recs=[{k: (v.cpu() if torch.is_tensor(v) else v) for k, v in r.items()} for r in recs])
    torch.save(res, f"{out}.{rank}")
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


# world 4 / 8: north_star shards num_envs over the EIGHT GPUs of a node.  The goldens hold 6 - 8 envs, so at world 8 every rank owns one env
# (mappo_dense, ippo_dense: B = 8) or some ranks own NONE (mappo_ragged_norm B = 7, ippo_ragged_norm B = 6, with all three normalisations
# on: the zero-count moment triples go through merge_moments_, the all-zero [gradient | statistics] buffers through every all-reduce and
# step -- learner.PPOLearner._empty_shard).  All ranks share cuda:0 (gloo carries the collectives: RCCL refuses two ranks on one device).
@pytest.mark.parametrize("world", [2, 4, 8])
@pytest.mark.parametrize("name,algo", [("mappo_ragged_norm", "mappo"), ("ippo_ragged_norm", "ippo"), ("mappo_dense", "mappo"), ("ippo_dense", "ippo")])
def test_env_shards_reproduce_the_single_process_reference(golden_dir, tmp_path, name, algo, world):
    if world == 4 and name in ("ippo_ragged_norm", "mappo_dense"):
        pytest.skip("world 4 is covered by the other two goldens")
    port, out = _free_port(), str(tmp_path / "rank")
    gold = os.path.join(golden_dir, name + ".npz")
    mp.spawn(_worker, args=(world, port, gold, algo, out), nprocs=world, join=True)
    z = np.load(gold)
    got = [torch.load(f"{out}.{r}", weights_only=False) for r in range(world)]
    kind = str(z["hp_optimizer"])
    for g in got:
        sl = slice(g["lo"], g["lo"] + g["n"])
        assert _err(g["ret"].numpy(), z["return_lambda"][sl]) <= TOL
        assert _err(g["adv"].numpy(), z["advantages"][sl]) <= TOL
        ca = StepChecker(golden_init(z, "actor"), kind, float(z["hp_learning_rate_actor"]), "env shards actor")
        cc = StepChecker(golden_init(z, "critic"), kind, float(z["hp_learning_rate_critic"]), "env shards critic")
        for e, r in enumerate(g["recs"]):
            assert _err(r["actor_loss"], z["actor_losses"][e]) <= TOL
            assert _err(r["critic_loss"], z["critic_losses"][e]) <= TOL
            assert _err(r["entropy"], z["entropies_bonuses"][e]) <= TOL
            assert _err(r["kl"], z["kl_divergences"][e]) <= TOL
            assert _err(r["clipfrac"], z["clipped_ratios"][e]) <= TOL
            assert grad_err(r["actor_gnorm"], z["actor_gradients"][e]) <= GRAD_TOL
            assert grad_err(r["critic_gnorm"], z["critic_gradients"][e]) <= GRAD_TOL
            ca.step(r["actor_grads"], r["actor_after"], z["actor_grads"][e], z["actor_after"][e])
            cc.step(r["critic_grads"], r["critic_after"], z["critic_grads"][e], z["critic_after"][e])
    assert sum(g["n"] for g in got) == z["b_obs"].shape[0] and (world <= z["b_obs"].shape[0] or any(g["n"] == 0 for g in got))
    # replicated parameters stay bit-identical across ranks (same reduced buffer, same Adam kernel) -- ranks without envs included
    for e in range(len(got[0]["recs"])):
        for r in range(1, world):
            assert torch.equal(got[0]["recs"][e]["actor_after"], got[r]["recs"][e]["actor_after"])
            assert torch.equal(got[0]["recs"][e]["critic_after"], got[r]["recs"][e]["critic_after"])


def _worker_peer(rank, world, port, gold, algo, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    os.environ["CM_PEER_ALLREDUCE"] = "1"
    torch.cuda.set_device(0)
    torch.distributed.init_process_group("gloo", rank=rank, world_size=world)  # control plane only: the gradients never touch it
    from oracle import restatement as R
    from cleanmarl_amd import dist
    from cleanmarl_amd.learner import DeviceBatch, HParams, NetSpec, PPOLearner
    batch, ap, cp, hp, z = R.load_golden(gold)
    dev = torch.device("cuda:0")
    reward = torch.from_numpy(z["b_reward_raw"]) if "b_reward_raw" in z.files else batch["reward"]
    lo, n = dist.shard(batch["obs"].shape[0], rank, world)
    sl = slice(lo, lo + n)
    H = HParams(gamma=hp["gamma"], td_lambda=hp["td_lambda"], normalize_reward=bool(hp["normalize_reward"]),
                normalize_advantage=bool(hp["normalize_advantage"]), normalize_return=bool(hp["normalize_return"]),
                epochs=int(hp["epochs"]), ppo_clip=hp["ppo_clip"], entropy_coef=hp["entropy_coef"],
                clip_gradients=hp["clip_gradients"], optimizer=hp["optimizer"],
                learning_rate_actor=hp["learning_rate_actor"], learning_rate_critic=hp["learning_rate_critic"])
    aspec = NetSpec(ap[0].shape[1], ap[0].shape[0], len(ap) // 2 - 2, ap[-1].shape[0])
    cspec = NetSpec(cp[0].shape[1], cp[0].shape[0], len(cp) // 2 - 2, 1)
    res = {}
    for sched in ("0", "1", "2"):
        os.environ["CM_CRITIC_OVERLAP"] = sched
        b = DeviceBatch.from_reference_layout(batch["obs"][sl], batch["actions"][sl], batch["log_probs"][sl], reward[sl],
                                              batch["states"][sl], batch["avail"][sl], batch["mask"][sl], dev, pad=True)
        L = PPOLearner(algo, aspec, cspec, batch["obs"].shape[2], H, dev, actor_params=[p.clone() for p in ap],
                       critic_params=[p.clone() for p in cp], process_group=torch.distributed.group.WORLD, world_size=world)
        assert L.peer_a is not None and L.peer_c is not None
        recs = [dict(r) for r in L.train_iteration(b)]
        more = [dict(r) for r in L.train_iteration(b)]  # a second iteration: the slot sets alternate, the tags keep counting
        L.wait_critic()
        torch.cuda.synchronize()
        res[sched] = dict(recs=recs, more=more, actor=L.actor.cpu(), critic=L.critic_params().cpu(), seq=(L.peer_a.seq, L.peer_c.seq))
        L.peer_a.close(); L.peer_c.close()
    torch.save(res, f"{out}.{rank}")
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4, 8])
@pytest.mark.parametrize("name,algo", [("mappo_ragged_norm", "mappo"), ("ippo_dense", "ippo")])
def test_peer_allreduce_processes_share_one_gpu(golden_dir, tmp_path, name, algo, world):
    """The one-shot peer all-reduce (csrc/cm_peer.hip; opt-in CM_PEER_ALLREDUCE=1): two processes on cuda:0 map each other's mailboxes
    with hipIpc, push their halves of the reference batch's gradient sums and step from the slots in rank order -- no all-reduce call on
    the data path.  The FIRST iteration must match the unmodified single-process reference per epoch, both ranks must hold bit-identical
    parameters after two iterations, for the one-stream and both two-stream schedules (actor and critic mailboxes on two streams).
    world 4 / 8: k_peer_push with gridDim.y = world, the [2][world][n] slot layout and the step kernel waiting on `world` tag words;
    mappo_ragged_norm has 7 envs, so at world 8 one rank pushes all-zero buffers (it owns no envs) and still steps."""
    port, out = _free_port(), str(tmp_path / "peer")
    gold = os.path.join(golden_dir, name + ".npz")
    mp.spawn(_worker_peer, args=(world, port, gold, algo, out), nprocs=world, join=True)
    z = np.load(gold)
    got = [torch.load(f"{out}.{r}", weights_only=False) for r in range(world)]
    nE = len(z["actor_losses"])
    for g in got:
        for sched, r in g.items():
            assert r["seq"] == (2 * nE, 2 * nE)
            for e, rec in enumerate(r["recs"]):
                assert _err(rec["actor_loss"], z["actor_losses"][e]) <= TOL and _err(rec["critic_loss"], z["critic_losses"][e]) <= TOL, sched
                assert grad_err(rec["actor_gnorm"], z["actor_gradients"][e]) <= GRAD_TOL and grad_err(rec["critic_gnorm"], z["critic_gradients"][e]) <= GRAD_TOL, sched
                assert _err(rec["entropy"], z["entropies_bonuses"][e]) <= TOL and _err(rec["kl"], z["kl_divergences"][e]) <= TOL, sched
    for sched in ("0", "1", "2"):
        for r in range(1, world):
            assert torch.equal(got[0][sched]["actor"], got[r][sched]["actor"]) and torch.equal(got[0][sched]["critic"], got[r][sched]["critic"]), (sched, r)
        assert torch.equal(got[0]["0"]["actor"], got[0][sched]["actor"]) and torch.equal(got[0]["0"]["critic"], got[0][sched]["critic"]), sched


def test_bench_two_rank_launch_on_one_gpu():
    """bench.py's N > 1 branch (rendezvous, env_offset sharding, barriers, MAX-over-ranks timing, rank-0 JSON line)
    launched exactly as the driver launches it, but over gloo with both ranks on cuda:0 (CM_BENCH_BACKEND test hook)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, CM_BENCH_BACKEND="gloo")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
           "--workload", "cfg2", "--envs", "96"]
    p = subprocess.run(cmd, env=env, cwd=root, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    out = json.loads(lines[0])
    # strong scaling is the default (north_star: num_envs shards across the GPUs): the 96 envs are GLOBAL, 48 per rank
    assert out["n_gpus"] == 2 and out["steps"] == 2 and out["warmup"] == 1 and out["scaling"] == "strong"
    assert out["config"]["global_envs"] == 96 and out["config"]["envs_per_gpu"] == 48
    assert out["value"] > 0 and abs(out["value"] - 96 * 3 * 128 * 2 / (out["ms_per_step"] * 2e-3)) < 1e-6 * out["value"]
    assert "cpu_baseline" not in out  # rank 0 at N = 1 only
    assert out["roofline"]["frac"] > 0
    # extras at N > 1: the two all-reduce messages timed alone (in the line, which is printed BEFORE the next leg so that leg can never
    # cost the headline record) and a short weak-scaling run (96 envs on EVERY rank) reported on stderr, tagged
    # both exchanges of an optimiser step timed alone: RCCL(here: gloo) all-reduce and the stand-alone step launch behind it, and the one-shot
    # peer exchange fused with the step (hipIpc mailboxes between the two processes on this one GPU), per message; + the exposure in the iteration
    lat = out["allreduce_us"]
    assert set(lat) == {"rccl", "optimizer_step", "peer_exchange_plus_step"}, lat
    for kind in lat.values():
        assert len(kind) == 2 and all(isinstance(v, float) and v > 0 for v in kind.values()), lat
    assert out["config"]["allreduce"] == "rccl" and "actor_stream_per_epoch_outside_the_pass" in out["comm_exposure_us"]
    assert "weak_scaling" not in out
    legs = [json.loads(ln.split("[bench extra leg] ", 1)[1]) for ln in p.stderr.splitlines() if "[bench extra leg] " in ln]
    assert len(legs) == 1 and legs[0]["leg"] == "weak_scaling" and legs[0]["n_gpus"] == 2
    wk = legs[0]
    assert wk["global_envs"] == 192 and abs(wk["value"] - 192 * 3 * 128 / (wk["ms_per_step"] * 1e-3)) < 1e-6 * wk["value"]
    # the same run with the peer exchange selected (rank-invariant flag): same work, same record shape
    p = subprocess.run(cmd + ["--allreduce", "peer", "--no-extras"], env=env, cwd=root, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    out_p = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][0])
    assert out_p["config"]["allreduce"] == "peer" and out_p["value"] > 0 and out_p["config"]["global_envs"] == 96
    cmd[cmd.index("--envs") + 1:cmd.index("--envs") + 2] = ["96", "--scaling", "weak", "--no-extras"]
    p = subprocess.run(cmd, env=env, cwd=root, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    out = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][0])
    assert out["scaling"] == "weak" and out["config"]["global_envs"] == 192 and "strong_scaling" not in out
    assert abs(out["value"] - 2 * 96 * 3 * 128 * 2 / (out["ms_per_step"] * 2e-3)) < 1e-6 * out["value"]


def test_bench_eight_rank_launch_on_one_gpu():
    """The driver's N = 8 command line (the node north_star shards over) with every extra leg, all eight ranks on cuda:0 over gloo:
    dist.shard deals 100 envs as 13 / 13 / 13 / 13 / 12 / 12 / 12 / 12, the message-latency leg builds EIGHT-rank peer mailboxes
    (k_peer_push with gridDim.y = 8, the step kernel waiting on eight tags), the weak leg runs 100 envs on every rank."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, CM_BENCH_BACKEND="gloo")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(root, "bench.py"), "--gpus", "8", "--steps", "2", "--warmup", "1",
           "--workload", "cfg2", "--envs", "100"]
    p = subprocess.run(cmd, env=env, cwd=root, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 8 and out["scaling"] == "strong" and out["config"]["global_envs"] == 100 and out["config"]["envs_per_gpu"] == 13
    assert out["value"] > 0 and abs(out["value"] - 100 * 3 * 128 * 2 / (out["ms_per_step"] * 2e-3)) < 1e-6 * out["value"]
    lat = out["allreduce_us"]
    assert all(isinstance(v, float) and v > 0 for kind in lat.values() for v in kind.values()), lat  # the peer exchange ran at world 8
    legs = [json.loads(ln.split("[bench extra leg] ", 1)[1]) for ln in p.stderr.splitlines() if "[bench extra leg] " in ln]
    assert len(legs) == 1 and legs[0]["leg"] == "weak_scaling" and legs[0]["global_envs"] == 800
    # the peer exchange as the data path of the timed region
    p = subprocess.run(cmd + ["--allreduce", "peer", "--no-extras"], env=env, cwd=root, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    out_p = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][0])
    assert out_p["config"]["allreduce"] == "peer" and out_p["value"] > 0 and out_p["n_gpus"] == 8


def test_peer_step_that_never_hears_from_a_peer_skips_the_update_and_reports_it():
    """the step kernel's wait for the peers' tags is bounded by WALL time; when it runs out the step is skipped -- parameters,
    optimiser state and the gradient buffer keep their values (no NaN poisoning) -- the logged norm is NaN and the sequence number appears
    in the caller's page-locked status word, which dist.PeerAllReduce.check() turns into a loud error.  One process plays rank 0 of a
    two-rank mailbox whose second rank never pushes."""
    import ctypes as C
    from cleanmarl_amd import _native as N
    lib = N.load()
    dev = torch.device("cuda:0")
    n, world = 1000, 2
    own, handle = C.c_void_p(), C.create_string_buffer(lib.cm_peer_handle_bytes())
    N.check(lib.cm_peer_mailbox_alloc(lib.cm_peer_mailbox_bytes(world, n + N.NUM_STATS), C.byref(own), handle), "alloc")
    try:
        boxes = (C.c_void_p * world)(own, own)  # "rank 1" is never pushed by anybody: only slot [1][0] gets its tag
        buf = torch.randn(n + N.NUM_STATS, device=dev)
        buf[n + N.STAT_COUNT] = 4.0
        params, m, v = torch.randn(n, device=dev), torch.rand(n, device=dev), torch.rand(n, device=dev)
        before = [t.clone() for t in (buf, params, m, v)]
        norm = torch.zeros(1, device=dev)
        scratch = torch.zeros(lib.cm_opt_step_scratch_bytes(), dtype=torch.uint8, device=dev)
        status = torch.zeros(1, dtype=torch.int32).pin_memory()
        o = N.OptStep(params=params.data_ptr(), exp_avg=m.data_ptr(), exp_avg_sq=v.data_ptr(), out_norm=norm.data_ptr(), scratch=scratch.data_ptr(),
                      lr=1e-2, beta1=0.9, beta2=0.999, eps=1e-8, weight_decay=0.0, max_norm=-1.0, grad_scale=1.0, step=1, opt_kind=N.OPT_ADAM)
        s = N.stream_ptr()
        N.check(lib.cm_peer_push(N.ptr(buf), n + N.NUM_STATS, 0, world, boxes, 1, s), "push")
        import time
        t0 = time.perf_counter()
        N.check(lib.cm_optimizer_step_peer(N.ptr(buf), n, own, world, 1, o, 0.25, N.ptr(status), s), "step")
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        assert 0.2 < dt < 5.0, dt  # the bound is wall time, not a poll count
        assert int(status[0]) == 1 and torch.isnan(norm).all()
        for t, b in zip((buf, params, m, v), before):
            assert torch.equal(t, b)
        # with clipping on (two launches: norm, then the clipped update) the second launch must skip as well
        o.max_norm = 0.5
        o.step = 2
        status.zero_()
        N.check(lib.cm_peer_push(N.ptr(buf), n + N.NUM_STATS, 0, world, boxes, 2, s), "push")
        N.check(lib.cm_optimizer_step_peer(N.ptr(buf), n, own, world, 2, o, 0.25, N.ptr(status), s), "step")
        torch.cuda.synchronize()
        assert int(status[0]) == 2
        for t, b in zip((params, m, v), before[1:]):
            assert torch.equal(t, b)
        # a healthy exchange (this process plays both ranks) on the same mailbox still steps: seq 3 into slot set 1
        status.zero_()
        o.max_norm = -1.0
        o.step = 1
        N.check(lib.cm_peer_push(N.ptr(buf), n + N.NUM_STATS, 0, world, boxes, 3, s), "push")
        N.check(lib.cm_peer_push(N.ptr(buf), n + N.NUM_STATS, 1, world, boxes, 3, s), "push")
        N.check(lib.cm_optimizer_step_peer(N.ptr(buf), n, own, world, 3, o, 0.25, N.ptr(status), s), "step")
        torch.cuda.synchronize()
        assert int(status[0]) == 0 and torch.isfinite(norm).all() and not torch.equal(params, before[1])
    finally:
        torch.cuda.synchronize()
        lib.cm_peer_mailbox_free(own)


def test_bench_starts_its_own_ranks_without_torchrun():
    """A bare `python bench.py --gpus 2` (no torch.distributed.run, WORLD_SIZE unset) must not die on the launcher check: it starts
    its ranks itself and prints rank 0's line (gloo test hook: both ranks on cuda:0)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env["CM_BENCH_BACKEND"] = "gloo"
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--workload", "cfg2",
                        "--envs", "96", "--no-extras"], env=env, cwd=root, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["config"]["global_envs"] == 96 and out["config"]["envs_per_gpu"] == 48 and out["value"] > 0


def _coma_worker(rank, world, port, gold, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.distributed.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import coma as C
    from cleanmarl_amd import dist
    from cleanmarl_amd.coma_learner import COMAHParams, COMALearner
    from cleanmarl_amd.learner import DeviceBatch, NetSpec
    batch, ap, cp, hp, z = C.load_golden(gold)
    dev = torch.device("cuda:0")
    lo, n = dist.shard(batch["obs"].shape[0], rank, world)
    sl = slice(lo, lo + n)
    b = DeviceBatch.from_reference_layout(batch["obs"][sl], batch["actions"][sl], torch.zeros(batch["actions"][sl].shape), batch["reward"][sl],
                                          batch["states"][sl], batch["avail"][sl], batch["mask"][sl], dev)
    H = COMAHParams(gamma=hp["gamma"], td_lambda=hp["td_lambda"], normalize_reward=False, normalize_advantage=bool(hp["normalize_advantage"]),
                    normalize_return=bool(hp["normalize_return"]), target_network_update_freq=int(hp["target_network_update_freq"]),
                    polyak=hp["polyak"], entropy_coef=hp["entropy_coef"], use_tdlamda=bool(hp["use_tdlamda"]), nsteps=int(hp["nsteps"]),
                    clip_gradients=hp["clip_gradients"], optimizer=hp["optimizer"], learning_rate_actor=hp["learning_rate_actor"],
                    learning_rate_critic=hp["learning_rate_critic"])
    aspec = NetSpec(ap[0].shape[1], ap[0].shape[0], len(ap) // 2 - 2, ap[-1].shape[0])
    cspec = NetSpec(cp[0].shape[1], cp[0].shape[0], len(cp) // 2 - 2, cp[-1].shape[0])
    L = COMALearner(aspec, cspec, batch["obs"].shape[2], H, dev, actor_params=ap, critic_params=cp,
                    process_group=torch.distributed.group.WORLD, world_size=world)
    rec = L.train_iteration(b)
    torch.save(dict(rec=rec, actor=L.actor.cpu(), critic=L.critic_params().cpu(), target=L.target.cpu()), f"{out}.{rank}")
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


@pytest.mark.parametrize("world", [2, 8])
def test_env_shards_reproduce_the_single_process_coma_reference(golden_dir, tmp_path, world):
    """COMA: per-time-step advantage moments and the return normalisation need cross-rank sums; the env shards must land on
    the unmodified single-process coma_multienvs.py result (tests/golden/coma_tdlambda.npz: ragged, normalisations on, clip).
    world 8: the golden holds 6 envs, so two ranks own none (COMALearner._empty_shard: zero buffers, every collective and step)."""
    port, out = _free_port(), str(tmp_path / "rank")
    gold = os.path.join(golden_dir, "coma_tdlambda.npz")
    mp.spawn(_coma_worker, args=(world, port, gold, out), nprocs=world, join=True)
    z = np.load(gold)
    got = [torch.load(f"{out}.{r}", weights_only=False) for r in range(world)]
    for g in got:
        assert _err(g["rec"]["critic_loss"], float(z["cr_loss"])) <= TOL and _err(g["rec"]["actor_loss"], float(z["ac_loss"])) <= TOL
        assert _err(g["rec"]["entropy"], float(z["entropies"])) <= TOL
        assert grad_err(g["rec"]["critic_gnorm"], float(z["critic_gradients"])) <= GRAD_TOL
        assert grad_err(g["rec"]["actor_gnorm"], float(z["actor_gradients"])) <= GRAD_TOL
        _check_final_params(g, z, last=False)
        assert _err(g["target"].numpy(), z["target_after"]) <= 1e-6
    for r in range(1, world):
        assert torch.equal(got[0]["actor"], got[r]["actor"]) and torch.equal(got[0]["critic"], got[r]["critic"])


def _gru_worker(rank, world, port, gold, algo, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.distributed.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import restatement as R
    from cleanmarl_amd import dist
    from cleanmarl_amd.gru import GRUPPOLearner
    from cleanmarl_amd.learner import DeviceBatch, HParams, NetSpec
    batch, ap, cp, hp, z = R.load_golden(gold)
    dev = torch.device("cuda:0")
    lo, n = dist.shard(batch["obs"].shape[0], rank, world)
    sl = slice(lo, lo + n)
    b = DeviceBatch.from_reference_layout(batch["obs"][sl], batch["actions"][sl], batch["log_probs"][sl], batch["reward"][sl],
                                          batch["states"][sl], batch["avail"][sl], batch["mask"][sl], dev)
    H = HParams(gamma=hp["gamma"], td_lambda=hp["td_lambda"], normalize_reward=False, normalize_advantage=bool(hp["normalize_advantage"]),
                normalize_return=bool(hp["normalize_return"]), epochs=int(hp["epochs"]), ppo_clip=hp["ppo_clip"],
                entropy_coef=hp["entropy_coef"], clip_gradients=hp["clip_gradients"], optimizer=hp["optimizer"], tbptt=int(hp["tbptt"]),
                learning_rate_actor=hp["learning_rate_actor"], learning_rate_critic=hp["learning_rate_critic"])
    aspec = NetSpec(ap[0].shape[1], ap[0].shape[0], 0, ap[-1].shape[0], "gru")
    cspec = NetSpec(cp[0].shape[1], cp[0].shape[0], len(cp) // 2 - 2, 1)
    L = GRUPPOLearner(algo, aspec, cspec, batch["obs"].shape[2], H, dev, actor_params=ap, critic_params=cp,
                      process_group=torch.distributed.group.WORLD, world_size=world)
    recs = L.train_iteration(b)
    torch.save(dict(recs=recs, actor=L.actor.cpu(), critic=L.critic_params().cpu()), f"{out}.{rank}")
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


@pytest.mark.parametrize("world", [2, 8])
@pytest.mark.parametrize("name,algo", [("mappo_lstm_ragged", "mappo"), ("ippo_lstm_ragged", "ippo")])
def test_env_shards_reproduce_the_single_process_gru_reference(golden_dir, tmp_path, name, algo, world):
    """GRU / TBPTT: one all-reduce per chunk (actor) + one per epoch (critic); the env shards must land on the post-update
    parameters of the unmodified single-process *_lstm_multienvs.py after all epochs x chunks optimiser steps.  world 8: the goldens
    hold 6 / 5 envs -- two / three ranks own none and still enter every chunk's all-reduce and step (GRUPPOLearner: _empty_shard)."""
    port, out = _free_port(), str(tmp_path / "rank")
    gold = os.path.join(golden_dir, name + ".npz")
    mp.spawn(_gru_worker, args=(world, port, gold, algo, out), nprocs=world, join=True)
    z = np.load(gold)
    got = [torch.load(f"{out}.{r}", weights_only=False) for r in range(world)]
    for g in got:
        for e, r in enumerate(g["recs"]):
            assert _err(r["actor_loss"], z["actor_losses"][e]) <= TOL and _err(r["critic_loss"], z["critic_losses"][e]) <= TOL
            assert _err(r["entropy"], z["entropies_bonuses"][e]) <= TOL and _err(r["kl"], z["kl_divergences"][e]) <= TOL
            assert grad_err(r["actor_gnorm"], z["actor_gradients"][e]) <= GRAD_TOL and grad_err(r["critic_gnorm"], z["critic_gradients"][e]) <= GRAD_TOL
        _check_final_params(g, z, last=True)
    for r in range(1, world):
        assert torch.equal(got[0]["actor"], got[r]["actor"]) and torch.equal(got[0]["critic"], got[r]["critic"])


def test_bench_contract_single_gpu():
    """The one JSON line `python bench.py` prints at N = 1: every field of the driver's contract, the roofline object of the
    dominant kernel and the bounded CPU baseline (a smaller sample here to keep the test short)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "2", "--warmup", "1", "--cpu-envs", "4"],
                       cwd=root, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in out, k
    assert out["n_gpus"] == 1 and out["steps"] == 2 and out["warmup"] == 1 and out["higher_is_better"] is True
    assert out["vs_baseline"] is None and out["dtype"] == "f32" and out["data"] == "synthetic" and out["scaling"] == "strong"
    assert "4096 envs x 8 agents x 128 steps" in out["config"]["workload"] and "model" not in out["config"]
    assert abs(out["value"] - 4096 * 8 * 128 / (out["ms_per_step"] * 1e-3)) < 1e-6 * out["value"]
    rf = out["roofline"]
    assert rf["bound"] == "mfma" and rf["unit"] == "TFLOP/s" and rf["peak"] == 157.3
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-9 and 0.3 < rf["frac"] < 1.0
    assert rf["traffic"] is None or rf["traffic"] > 0.9 * rf["algorithmic_bytes_per_launch"]
    # the kernel's own clock probe: the shader clock of the last actor pass, its rate against the peak AT that clock, ramp and tail
    assert 1.5 < rf["shader_clock_ghz"] < 2.6 and rf["frac_of_peak_at_shader_clock"] >= 0.95 * rf["frac"]
    ws = rf["workgroup_span"]
    assert ws["workgroups"] == 512 and ws["last_entry_ms"] < 0.05 < ws["first_exit_ms"] <= ws["median_exit_ms"] <= ws["last_exit_ms"] < rf["workgroup_span"]["launch_ms_hip_events"]
    cb = out["cpu_baseline"]
    assert cb["kind"] == "port" and cb["value"] > 0 and cb["cores"] >= 1 and "envs" in cb["sample"]
    assert out["value"] > 50 * cb["value"]  # north-star: >= 50x the reference-structured CPU path
    # every BASELINE config is in the driver-run line, and the per-GPU shares of the sharded ones
    assert set(out["other_workloads"]) == {"cfg2", "cfg4", "cfg5"}
    assert all(v["ms_per_step"] > 0 and 0 < v["roofline_frac"] < 1 for v in out["other_workloads"].values())
    assert set(out["strong_scaling_shares"]) == {"cfg3", "cfg4"}
    sh = out["strong_scaling_shares"]["cfg3"]["shares"]["1/8 (512 envs)"]
    assert 1.0 < sh["speedup_bound"] < 8.5
    pj = out["projected_speedup_8"]
    assert pj["exposed_messages_per_iteration"] == 3 and pj["latency_measured"] is True and 5.0 < pj["latency_us"] < 500.0
    assert "one-rank RCCL" in pj["latency_source"] and pj["at_30us"] < pj["without_communication"]
    assert abs(pj["value"] - pj["full_ms"] / (pj["share_ms"] + 3e-3 * pj["latency_us"])) < 1e-9 and pj["value"] < pj["without_communication"] == sh["speedup_bound"]
    pr = out["phase_roofline"]
    assert set(pr) == {"rollout", "value_pass_scan", "critic_fwd_bwd", "whole_step"} and all(0 < v["frac"] < 1 for v in pr.values())


def _worker_overlap(rank, world, port, gold, algo, out):
    """Same as _worker but WITHOUT keep_grads: update() then takes the two-stream schedule (critic epochs on the low-priority stream, own
    communicator, not joined) that small batches use in production."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.distributed.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import restatement as R
    from cleanmarl_amd import dist
    from cleanmarl_amd.learner import DeviceBatch, HParams, NetSpec, PPOLearner
    batch, ap, cp, hp, z = R.load_golden(gold)
    dev = torch.device("cuda:0")
    reward = torch.from_numpy(z["b_reward_raw"]) if "b_reward_raw" in z.files else batch["reward"]
    lo, n = dist.shard(batch["obs"].shape[0], rank, world)
    sl = slice(lo, lo + n)
    b = DeviceBatch.from_reference_layout(batch["obs"][sl], batch["actions"][sl], batch["log_probs"][sl], reward[sl],
                                          batch["states"][sl], batch["avail"][sl], batch["mask"][sl], dev, pad=True)
    H = HParams(gamma=hp["gamma"], td_lambda=hp["td_lambda"], normalize_reward=bool(hp["normalize_reward"]),
                normalize_advantage=bool(hp["normalize_advantage"]), normalize_return=bool(hp["normalize_return"]),
                epochs=int(hp["epochs"]), ppo_clip=hp["ppo_clip"], entropy_coef=hp["entropy_coef"],
                clip_gradients=hp["clip_gradients"], optimizer=hp["optimizer"],
                learning_rate_actor=hp["learning_rate_actor"], learning_rate_critic=hp["learning_rate_critic"])
    aspec = NetSpec(ap[0].shape[1], ap[0].shape[0], len(ap) // 2 - 2, ap[-1].shape[0])
    cspec = NetSpec(cp[0].shape[1], cp[0].shape[0], len(cp) // 2 - 2, 1)
    L = PPOLearner(algo, aspec, cspec, batch["obs"].shape[2], H, dev, actor_params=ap, critic_params=cp,
                   process_group=torch.distributed.group.WORLD, world_size=world)
    assert L.overlap_critic(b) in (1, 2) and L.pg_c is not L.pg   # 1 at this size by default; the parametrised run forces each
    recs = L.train_iteration(b)
    recs = [dict(r) for r in recs]  # materialises the lazily copied statistics (waits for both streams)
    L.wait_critic()
    torch.cuda.synchronize()
    torch.save(dict(recs=recs, actor=L.actor.cpu(), critic=L.critic_params().cpu()), f"{out}.{rank}")
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


@pytest.mark.parametrize("name,algo", [("mappo_ragged_norm", "mappo"), ("ippo_ragged_norm", "ippo")])
def test_two_ranks_two_stream_schedule_reproduces_the_reference(golden_dir, tmp_path, name, algo):
    """The production schedule of small batches at N = 2 (critic epochs on their own low-priority stream and communicator, released with the
    actor's, joined by the next reader; padded leading dimensions): final parameters and per-epoch scalars of the unmodified single-process
    reference, bit-identical across ranks."""
    world, out = 2, str(tmp_path / "rank")
    gold = os.path.join(golden_dir, name + ".npz")
    for sched in ("1", "2"):  # critic epochs released after / with the actor's
        os.environ["CM_CRITIC_OVERLAP"] = sched
        try:
            mp.spawn(_worker_overlap, args=(world, _free_port(), gold, algo, out), nprocs=world, join=True)
        finally:
            del os.environ["CM_CRITIC_OVERLAP"]
        _check_overlap_ranks(gold, out, world)


def _check_overlap_ranks(gold, out, world):
    z = np.load(gold)
    got = [torch.load(f"{out}.{r}", weights_only=False) for r in range(world)]
    for g in got:
        _check_final_params(g, z, last=True)
        for e, r in enumerate(g["recs"]):
            assert _err(r["actor_loss"], z["actor_losses"][e]) <= TOL and _err(r["critic_loss"], z["critic_losses"][e]) <= TOL
            assert grad_err(r["actor_gnorm"], z["actor_gradients"][e]) <= GRAD_TOL and grad_err(r["critic_gnorm"], z["critic_gradients"][e]) <= GRAD_TOL
    assert torch.equal(got[0]["actor"], got[1]["actor"]) and torch.equal(got[0]["critic"], got[1]["critic"])


def _worker_rccl(rank, world, port, gold, algo, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    torch.cuda.set_device(0)
    torch.distributed.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda:0"))
    from oracle import restatement as R
    from cleanmarl_amd.learner import DeviceBatch, HParams, NetSpec, PPOLearner
    batch, ap, cp, hp, z = R.load_golden(gold)
    dev = torch.device("cuda:0")
    reward = torch.from_numpy(z["b_reward_raw"]) if "b_reward_raw" in z.files else batch["reward"]
    H = HParams(gamma=hp["gamma"], td_lambda=hp["td_lambda"], normalize_reward=bool(hp["normalize_reward"]),
                normalize_advantage=bool(hp["normalize_advantage"]), normalize_return=bool(hp["normalize_return"]),
                epochs=int(hp["epochs"]), ppo_clip=hp["ppo_clip"], entropy_coef=hp["entropy_coef"],
                clip_gradients=hp["clip_gradients"], optimizer=hp["optimizer"],
                learning_rate_actor=hp["learning_rate_actor"], learning_rate_critic=hp["learning_rate_critic"])
    aspec = NetSpec(ap[0].shape[1], ap[0].shape[0], len(ap) // 2 - 2, ap[-1].shape[0])
    cspec = NetSpec(cp[0].shape[1], cp[0].shape[0], len(cp) // 2 - 2, 1)
    res = {}
    for sched in ("0", "1", "2"):
        os.environ["CM_CRITIC_OVERLAP"] = sched
        b = DeviceBatch.from_reference_layout(batch["obs"], batch["actions"], batch["log_probs"], reward, batch["states"], batch["avail"],
                                              batch["mask"], dev, pad=True)
        L = PPOLearner(algo, aspec, cspec, batch["obs"].shape[2], H, dev, actor_params=[p.clone() for p in ap],
                       critic_params=[p.clone() for p in cp], process_group=torch.distributed.group.WORLD, world_size=world)
        assert L._coll and L.pg_c is not L.pg
        recs = [dict(r) for r in L.train_iteration(b)]
        L.wait_critic()
        torch.cuda.synchronize()
        res[sched] = dict(recs=recs, actor=L.actor.cpu(), critic=L.critic_params().cpu())
    torch.save(res, f"{out}.{rank}")
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


@pytest.mark.parametrize("name,algo", [("mappo_ragged_norm", "mappo")])
def test_rccl_carries_the_all_reduces_of_every_update_schedule(golden_dir, tmp_path, name, algo, monkeypatch):
    """The data-path collectives over RCCL itself (backend "nccl"), on the one GPU a test box has: world size 1 with CM_FORCE_COLLECTIVES=1
    makes update() issue every asynchronous all-reduce it would issue at N > 1 -- actor and critic messages on their own communicators, from
    the launch stream and the low-priority stream, waited for by stream order -- for the one-stream and both two-stream schedules.  A
    one-rank sum is the identity, so each schedule must land on the unmodified reference's post-update parameters; what this pins is that
    RCCL initialises, accepts the two communicators and the handles' stream-side waits, and neither deadlocks nor reorders the steps."""
    monkeypatch.setenv("CM_FORCE_COLLECTIVES", "1")
    out, gold = str(tmp_path / "rccl"), os.path.join(golden_dir, name + ".npz")
    mp.spawn(_worker_rccl, args=(1, _free_port(), gold, algo, out), nprocs=1, join=True)
    z = np.load(gold)
    got = torch.load(f"{out}.0", weights_only=False)
    for sched, g in got.items():
        _check_final_params(g, z, last=True)
        for e, r in enumerate(g["recs"]):
            assert _err(r["actor_loss"], z["actor_losses"][e]) <= TOL and _err(r["critic_loss"], z["critic_losses"][e]) <= TOL, sched
    assert torch.equal(got["1"]["actor"], got["2"]["actor"]) and torch.equal(got["1"]["critic"], got["2"]["critic"])


def _worker_rccl_gru_coma(rank, world, port, gold_gru, gold_coma, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    torch.cuda.set_device(0)
    torch.distributed.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda:0"))
    from oracle import coma as C
    from oracle import restatement as R
    from cleanmarl_amd.coma_learner import COMAHParams, COMALearner
    from cleanmarl_amd.gru import GRUPPOLearner
    from cleanmarl_amd.learner import DeviceBatch, HParams, NetSpec
    dev, pg = torch.device("cuda:0"), torch.distributed.group.WORLD
    res = {}
    # ---- GRU / TBPTT: one blocking all-reduce per chunk on the launch stream + the critic's on the side stream and its own communicator
    batch, ap, cp, hp, z = R.load_golden(gold_gru)
    b = DeviceBatch.from_reference_layout(batch["obs"], batch["actions"], batch["log_probs"], batch["reward"], batch["states"], batch["avail"],
                                          batch["mask"], dev)
    H = HParams(gamma=hp["gamma"], td_lambda=hp["td_lambda"], normalize_reward=False, normalize_advantage=bool(hp["normalize_advantage"]),
                normalize_return=bool(hp["normalize_return"]), epochs=int(hp["epochs"]), ppo_clip=hp["ppo_clip"], entropy_coef=hp["entropy_coef"],
                clip_gradients=hp["clip_gradients"], optimizer=hp["optimizer"], tbptt=int(hp["tbptt"]),
                learning_rate_actor=hp["learning_rate_actor"], learning_rate_critic=hp["learning_rate_critic"])
    aspec = NetSpec(ap[0].shape[1], ap[0].shape[0], 0, ap[-1].shape[0], "gru")
    cspec = NetSpec(cp[0].shape[1], cp[0].shape[0], len(cp) // 2 - 2, 1)
    L = GRUPPOLearner("mappo", aspec, cspec, batch["obs"].shape[2], H, dev, actor_params=ap, critic_params=cp, process_group=pg, world_size=world)
    assert L._coll and L.pg_c is not L.pg
    recs = [dict(r) for r in L.train_iteration(b)]
    torch.cuda.synchronize()
    res["gru"] = dict(recs=recs, actor=L.actor.cpu(), critic=L.critic_params().cpu())
    # ---- COMA: gradient buffers + the float64 per-time-step advantage sums
    batch, ap, cp, hp, z = C.load_golden(gold_coma)
    b = DeviceBatch.from_reference_layout(batch["obs"], batch["actions"], torch.zeros(batch["actions"].shape), batch["reward"], batch["states"],
                                          batch["avail"], batch["mask"], dev)
    Hc = COMAHParams(gamma=hp["gamma"], td_lambda=hp["td_lambda"], normalize_reward=False, normalize_advantage=bool(hp["normalize_advantage"]),
                     normalize_return=bool(hp["normalize_return"]), target_network_update_freq=int(hp["target_network_update_freq"]),
                     polyak=hp["polyak"], entropy_coef=hp["entropy_coef"], use_tdlamda=bool(hp["use_tdlamda"]), nsteps=int(hp["nsteps"]),
                     clip_gradients=hp["clip_gradients"], optimizer=hp["optimizer"], learning_rate_actor=hp["learning_rate_actor"],
                     learning_rate_critic=hp["learning_rate_critic"])
    aspec = NetSpec(ap[0].shape[1], ap[0].shape[0], len(ap) // 2 - 2, ap[-1].shape[0])
    cspec = NetSpec(cp[0].shape[1], cp[0].shape[0], len(cp) // 2 - 2, cp[-1].shape[0])
    Lc = COMALearner(aspec, cspec, batch["obs"].shape[2], Hc, dev, actor_params=ap, critic_params=cp, process_group=pg, world_size=world)
    assert Lc._coll
    rec = Lc.train_iteration(b)
    torch.cuda.synchronize()
    res["coma"] = dict(rec=rec, actor=Lc.actor.cpu(), critic=Lc.critic.cpu(), target=Lc.target.cpu())
    torch.save(res, f"{out}.{rank}")
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


def test_rccl_carries_the_all_reduces_of_the_gru_and_coma_learners(golden_dir, tmp_path, monkeypatch):
    """The other two learners over RCCL (one rank, CM_FORCE_COLLECTIVES=1): GRUPPOLearner's blocking per-chunk all-reduce on the launch
    stream with the critic's epoch on a second stream and communicator (gru.py), COMALearner's gradient and float64 moment
    all-reduces -- a pattern no NCCL-backed run had executed before round 3.  One-rank sums are identities: both must land on the
    unmodified reference's results."""
    monkeypatch.setenv("CM_FORCE_COLLECTIVES", "1")
    out = str(tmp_path / "rccl2")
    gg, gc = os.path.join(golden_dir, "mappo_lstm_ragged.npz"), os.path.join(golden_dir, "coma_tdlambda.npz")
    mp.spawn(_worker_rccl_gru_coma, args=(1, _free_port(), gg, gc, out), nprocs=1, join=True)
    got = torch.load(f"{out}.0", weights_only=False)
    z = np.load(gg)
    g = got["gru"]
    for e, r in enumerate(g["recs"]):
        assert _err(r["actor_loss"], z["actor_losses"][e]) <= TOL and _err(r["critic_loss"], z["critic_losses"][e]) <= TOL
        assert grad_err(r["actor_gnorm"], z["actor_gradients"][e]) <= GRAD_TOL and grad_err(r["critic_gnorm"], z["critic_gradients"][e]) <= GRAD_TOL
    _check_final_params(g, z, last=True)
    z = np.load(gc)
    c = got["coma"]
    assert _err(c["rec"]["critic_loss"], float(z["cr_loss"])) <= TOL and _err(c["rec"]["actor_loss"], float(z["ac_loss"])) <= TOL
    _check_final_params(c, z, last=False)
    assert _err(c["target"].numpy(), z["target_after"]) <= 1e-6


@pytest.mark.parametrize("script", ["mappo_multienvs", "mappo_lstm_multienvs", "coma_multienvs"])
def test_cli_run_over_rccl_equals_the_plain_run(script, tmp_path):
    """A whole CLI run (driver / coma_driver) with a one-rank RCCL process group and every collective issued (CM_FORCE_COLLECTIVES=1)
    logs the scalars of the plain single-process run of the same command."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    mod = "coma_driver" if script.startswith("coma") else "driver"
    prog = ("import json, sys; sys.path.insert(0, %r); from cleanmarl_amd.%s import run; "
            "out = run(%r, ['--env_type=synthetic', '--batch_size=6', '--synthetic_agents=3', '--synthetic_steps=12', "
            "'--total_timesteps=216', '--eval_steps=100000', '--log_every=1']); print('HIST ' + json.dumps(out['history']))") % (root, mod, script)
    path = str(tmp_path / "run_cli.py")
    open(path, "w").write(prog + "\n")
    hists = []
    for force in ("0", "1"):
        env = dict(os.environ, CM_FORCE_COLLECTIVES=force, MASTER_PORT=str(_free_port()), HSA_ENABLE_IPC_MODE_LEGACY="0")
        p = subprocess.run([sys.executable, path], cwd=str(tmp_path), env=env, capture_output=True, text=True, timeout=600)
        assert p.returncode == 0, p.stderr[-3000:]
        hists.append(json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("HIST ")][0][5:]))
    plain, forced = hists
    assert len(plain) == len(forced) and len(plain) > 0
    for (t1, v1, s1), (t2, v2, s2) in zip(plain, forced):
        assert t1 == t2 and s1 == s2
        if not t1.startswith("charts/") and "SPS" not in t1 and "time" not in t1:
            assert abs(v1 - v2) <= 1e-5 * (1 + abs(v1)), (t1, v1, v2)


def test_two_rank_cli_run_with_host_envs_matches_the_single_process_run(tmp_path):
    """mappo_multienvs.py launched as two ranks (torch.distributed.run, gloo test hook, both on cuda:0) with HOST envs: every rank owns the
    global env indices of its shard and its own Philox rows (all ranks used to collect the same episodes), the logged rollout
    statistics are those of ALL envs, and the first iteration's scalars equal the single-process run of the same command."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    prog = ("import json, sys; sys.path.insert(0, %r); from cleanmarl_amd.driver import run; "
            "out = run('mappo_multienvs', ['--env_type=synthetic_cpu', '--batch_size=6', '--synthetic_agents=3', '--synthetic_steps=9', "
            "'--total_timesteps=108', '--eval_steps=100000', '--log_every=1', '--vector_env=pinned']); "
            "import os; "
            "print('HIST ' + json.dumps(out['history'])) if os.environ.get('RANK', '0') == '0' else None") % root
    env = dict(os.environ, CM_DIST_BACKEND="gloo")
    script = str(tmp_path / "run_cli.py")
    open(script, "w").write(prog + "\n")
    one = subprocess.run([sys.executable, script], cwd=str(tmp_path), env=env, capture_output=True, text=True, timeout=600)
    assert one.returncode == 0, one.stderr[-3000:]
    two = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                          "--master-port", str(_free_port()), script], cwd=str(tmp_path), env=env, capture_output=True, text=True, timeout=600)
    assert two.returncode == 0, two.stderr[-3000:]
    hist = lambda p: json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("HIST ")][0][5:])
    h1, h2 = hist(one), hist(two)
    first = lambda h, tag: [v for t, v, s in h if t == tag][0]
    for tag in ("rollout/ep_reward", "rollout/ep_length", "rollout/num_episodes"):
        assert abs(first(h1, tag) - first(h2, tag)) <= 1e-5 * (1 + abs(first(h1, tag))), tag   # same episodes, all 6 envs in the mean
    for tag in ("train/actor_loss", "train/critic_loss", "train/entropy", "train/actor_gradients", "train/critic_gradients"):
        assert abs(first(h1, tag) - first(h2, tag)) <= 1e-4 * (1 + abs(first(h1, tag))), tag
    assert [s for t, v, s in h1 if t == "train/num_updates"] == [s for t, v, s in h2 if t == "train/num_updates"]


@pytest.mark.parametrize("world", [2, 8])
@pytest.mark.parametrize("env_type,script_name", [("synthetic_cpu", "mappo_multienvs"), ("synthetic", "mappo_multienvs"),
                                                  ("synthetic", "mappo_lstm_multienvs"), ("synthetic_cpu", "coma_multienvs")])
def test_multi_rank_evaluation_equals_the_single_process_evaluation(env_type, script_name, world, tmp_path):
    """Evaluation at N > 1 (cleanmarl_amd/evaluate.py): host envs -- the num_eval_ep episodes are dealt to the ranks in contiguous blocks and
    gathered (world 8 with 5 episodes: blocks of one, three ranks play nothing); device envs -- rank 0 plays them as one rollout on its
    evaluation stream while no rank waits.  The evaluators are compared ON IDENTICAL PARAMETERS: with both learning rates at 0 the
    optimiser steps (all issued, all-reduces included) leave the initial policy in place on every rank count, so the 1-rank and the
    N-rank run must log EXACTLY the same episode returns, round after round (greedy: no argmax can flip).  A second pair of
    runs with the default learning rates stays as a smoke check of the trained policy (same rounds, same lengths, finite returns)."""
    import json
    import subprocess
    import sys
    if world == 8 and script_name != "mappo_multienvs":
        pytest.skip("world 8 is covered by the MAPPO front-end (host and device evaluators)")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    mod = "coma_driver" if script_name.startswith("coma") else "driver"
    envs = 8 if world == 8 else 6
    hist = lambda p: json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("HIST ")][0][5:])
    ev = lambda h, tag: [(v, s) for t, v, s in h if t == tag]
    env = dict(os.environ, CM_DIST_BACKEND="gloo")
    for lrs in (["--learning_rate_actor=0", "--learning_rate_critic=0"], []):
        prog = ("import json, os, sys; sys.path.insert(0, %r); from cleanmarl_amd.%s import run; "
                "out = run(%r, ['--env_type=%s', '--batch_size=%d', '--synthetic_agents=3', '--synthetic_steps=9', "
                "'--total_timesteps=%d', '--eval_steps=1', '--num_eval_ep=5', '--log_every=1', '--greedy_eval'] + %r); "
                "print('HIST ' + json.dumps(out['history'])) if os.environ.get('RANK', '0') == '0' else None") % (
                    root, mod, script_name, env_type, envs, 2 * 9 * envs, lrs)
        script = str(tmp_path / "run_cli.py")
        open(script, "w").write(prog + "\n")
        one = subprocess.run([sys.executable, script], cwd=str(tmp_path), env=env, capture_output=True, text=True, timeout=600)
        assert one.returncode == 0, one.stderr[-3000:]
        many = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr",
                               "127.0.0.1", "--master-port", str(_free_port()), script], cwd=str(tmp_path), env=env, capture_output=True,
                              text=True, timeout=900)
        assert many.returncode == 0, many.stderr[-3000:]
        h1, h2 = hist(one), hist(many)
        for tag in ("eval/ep_reward", "eval/std_ep_reward", "eval/ep_length"):
            a, b = ev(h1, tag), ev(h2, tag)
            assert len(a) == len(b) >= 2 and [s for _, s in a] == [s for _, s in b], tag  # same evaluation rounds at the same env-step x values
            if lrs:  # identical parameters: identical episodes, bit for bit
                assert [v for v, _ in a] == [v for v, _ in b], (tag, a, b)
            else:
                assert all(np.isfinite(v) for v, _ in b), (tag, b)
        assert [v for v, _ in ev(h2, "eval/ep_length")] == [9.0] * len(ev(h2, "eval/ep_length"))
