"""GPU tests of the optimiser step fused behind the training passes (cm_*_train_step*, cm_optimizer_step; csrc/cm_optim.hip):
the single launch must leave EXACTLY the parameters, moments and gradients of the pass followed by the stand-alone
reduce / norm / update launches (cm_grad_norm_clip_adam) -- same summation order, same roundings -- for every schedule that
carries it (fused actor tile kernel, one-pass critic, two-kernel split critic with its two partial sets, GRU chunk sweeps,
layered fallback), with and without gradient clipping, for every optimiser kind.  The reference region is
cleanmarl/mappo_multienvs.py:572-594 (mappo_lstm_multienvs.py:605-618 for the chunk steps)."""
import numpy as np
import pytest
import torch

from test_hip_parity import _random_case

pytestmark = pytest.mark.gpu


def _mlp_learners(algo, E, A, T, Do, Ds, K, H, L, clip, optimizer, epochs=2):
    from cleanmarl_amd.learner import DeviceBatch, HParams, NetSpec, PPOLearner, init_params_like_torch
    torch.manual_seed(5)
    batch = _random_case(31, E, A, T, Do, Ds, K)
    aspec, cspec = NetSpec(Do, H, L, K), NetSpec(Ds if algo == "mappo" else Do, H, L, 1)
    ap, cp = init_params_like_torch(aspec), init_params_like_torch(cspec)
    hp = HParams(epochs=epochs, clip_gradients=clip, optimizer=optimizer, normalize_advantage=True, entropy_coef=0.01)
    dev = torch.device("cuda:0")
    out = []
    for fused in (True, False):
        b = DeviceBatch.from_reference_layout(batch["obs"], batch["actions"], batch["log_probs"], batch["reward"], batch["states"],
                                              batch["avail"], batch["mask"], dev, pad=True)
        Lr = PPOLearner(algo, aspec, cspec, A, hp, dev, actor_params=[p.clone() for p in ap], critic_params=[p.clone() for p in cp])
        Lr.fused_step = fused
        out.append((Lr, b))
    return out


def _same(x, y):
    return bool(torch.equal(x, y))


@pytest.mark.parametrize("clip", [-1.0, 0.5])
@pytest.mark.parametrize("optimizer", ["Adam", "AdamW", "SGD", "RMSprop"])
@pytest.mark.parametrize("sched", ["fused", "split"])
@pytest.mark.parametrize("algo,E,A,T,Do,Ds,K,H,L", [("mappo", 16, 8, 32, 56, 384, 5, 64, 1), ("ippo", 6, 3, 20, 140, 10, 5, 64, 1),
                                                      ("mappo", 37, 3, 25, 21, 54, 5, 32, 2),
                                                      # 512 partial rows for the actor (four row batches per group), 141 for the critic (a ragged last batch)
                                                      ("mappo", 300, 8, 30, 56, 384, 5, 64, 1)])
def test_fused_step_equals_the_three_launch_step_bit_for_bit(algo, E, A, T, Do, Ds, K, H, L, sched, optimizer, clip, monkeypatch):
    monkeypatch.setenv("CM_CRITIC_SCHEDULE", sched)  # one-pass critic where its shape rules allow / two-kernel split schedule (two partial sets)
    (Lf, bf), (Lu, bu) = _mlp_learners(algo, E, A, T, Do, Ds, K, H, L, clip, optimizer)
    rf, ru = Lf.train_iteration(bf, keep_grads=True), Lu.train_iteration(bu, keep_grads=True)
    torch.cuda.synchronize()
    # without clipping nothing depends on the norm: bit-identical.  With clipping the coefficient max_norm / (norm + 1e-6) inherits the
    # norm's summation order (per 64-column slab here, per thread stride in k_grad_norm_small): equal to a few ulp
    same = _same if clip <= 0 else (lambda x, y: bool(torch.allclose(x, y, rtol=1e-5, atol=1e-7)))
    assert same(Lf.actor, Lu.actor) and same(Lf.critic_params(), Lu.critic_params())
    for of, ou in ((Lf.opt_a, Lu.opt_a), (Lf.opt_c, Lu.opt_c)):
        assert of.step == ou.step and same(of.m, ou.m) and same(of.v, ou.v)
    for a, b in zip(rf, ru):
        assert same(a["actor_grads"], b["actor_grads"]) and same(a["critic_grads"], b["critic_grads"])
        for k in ("actor_loss", "critic_loss", "entropy", "kl", "clipfrac"):
            assert a[k] == b[k] if clip <= 0 else abs(a[k] - b[k]) <= 1e-6 * (1 + abs(b[k])), k
        for k in ("actor_gnorm", "critic_gnorm"):  # the norm is summed per 64-column slab instead of per thread stride
            assert abs(a[k] - b[k]) <= 2e-6 * (1 + abs(b[k])), (k, a[k], b[k])


def test_overlapped_schedules_carry_the_fused_step(monkeypatch):
    """The critic's epochs on their own stream (learner.overlap_critic schedules 1 and 2) ride on the same launches: three iterations
    of each schedule end on the parameters of the single-stream run."""
    from cleanmarl_amd.learner import PPOLearner  # noqa: F401
    ref = None
    for sched in ("0", "1", "2"):
        monkeypatch.setenv("CM_CRITIC_OVERLAP", sched)
        (Lf, bf), _ = _mlp_learners("mappo", 48, 3, 24, 21, 54, 5, 64, 1, -1.0, "Adam", epochs=3)
        for _ in range(3):
            Lf.train_iteration(bf)
        torch.cuda.synchronize()
        cur = (Lf.actor.clone(), Lf.critic_params().clone())
        if ref is None:
            ref = cur
        assert _same(cur[0], ref[0]) and _same(cur[1], ref[1]), sched


@pytest.mark.parametrize("clip", [-1.0, 0.5])
def test_gru_chunk_steps_ride_on_the_sweeps(clip):
    from cleanmarl_amd.gru import GRUPPOLearner
    from cleanmarl_amd.learner import DeviceBatch, HParams, NetSpec, init_params_like_torch
    torch.manual_seed(2)
    E, A, T, Do, Ds, K, H, tb = 40, 5, 23, 35, 150, 5, 64, 10
    batch = _random_case(77, E, A, T, Do, Ds, K)
    aspec, cspec = NetSpec(Do, H, 0, K, "gru"), NetSpec(Ds, 64, 1, 1)
    ap, cp = init_params_like_torch(aspec), init_params_like_torch(cspec)
    hp = HParams(epochs=2, clip_gradients=clip, tbptt=tb, normalize_advantage=True)
    dev = torch.device("cuda:0")
    res = []
    for fused in (True, False):
        b = DeviceBatch.from_reference_layout(batch["obs"], batch["actions"], batch["log_probs"], batch["reward"], batch["states"],
                                              batch["avail"], batch["mask"], dev)
        L = GRUPPOLearner("mappo", aspec, cspec, A, hp, dev, actor_params=[p.clone() for p in ap], critic_params=[p.clone() for p in cp])
        L.fused_step = fused
        recs = L.train_iteration(b, keep_grads=True)
        torch.cuda.synchronize()
        res.append((L, list(recs)))
    (Lf, rf), (Lu, ru) = res
    same = _same if clip <= 0 else (lambda x, y: bool(torch.allclose(x, y, rtol=1e-5, atol=1e-7)))
    assert same(Lf.actor, Lu.actor) and same(Lf.critic, Lu.critic) and same(Lf.opt_a.m, Lu.opt_a.m) and same(Lf.opt_a.v, Lu.opt_a.v)
    assert Lf.opt_a.step == Lu.opt_a.step == 2 * 3
    for a, b in zip(rf, ru):
        for (ga, pa), (gb, pb) in zip(a["actor_steps"], b["actor_steps"]):
            assert same(ga, gb) and same(pa, pb)
        assert abs(a["actor_gnorm"] - b["actor_gnorm"]) <= 2e-6 * (1 + abs(b["actor_gnorm"]))


def test_optimizer_step_matches_torch_adam_on_a_reduced_gradient():
    """cm_optimizer_step on an already reduced buffer (what follows an all-reduce) against torch.optim.Adam + clip_grad_norm_."""
    from cleanmarl_amd import _native as N
    from cleanmarl_amd.learner import _Adam
    lib, dev = N.load(), torch.device("cuda:0")
    for n, clip in ((8141, -1.0), (8141, 0.3), (28865, 0.3), (70001, -1.0)):  # the last one exceeds the fused launch's scratch: fallback path
        g = torch.Generator().manual_seed(n)
        p0, grads = torch.randn(n, generator=g), [torch.randn(n, generator=g) * 50 for _ in range(3)]
        count = 37.0
        pt = torch.nn.Parameter(p0.clone())
        topt = torch.optim.Adam([pt], lr=8e-4)
        p = p0.clone().to(dev)
        opt = _Adam(n, 8e-4, "Adam", dev)
        norm = torch.zeros(1, device=dev)
        for gr in grads:
            buf = torch.zeros(n + N.NUM_STATS, device=dev)
            buf[:n] = gr.to(dev)
            buf[n + N.STAT_COUNT] = count
            o = opt.next_step(p, norm, clip)
            N.check(lib.cm_optimizer_step(N.ptr(buf), n, o, N.stream_ptr()), "cm_optimizer_step")
            pt.grad = gr / count
            tn = torch.linalg.vector_norm(pt.grad).item()
            if clip > 0:
                torch.nn.utils.clip_grad_norm_([pt], clip)
            topt.step()
            torch.cuda.synchronize()
            assert abs(norm.item() - tn) <= 1e-5 * (1 + tn)
            np.testing.assert_allclose(buf[:n].cpu().numpy(), pt.grad.numpy(), rtol=1e-5, atol=1e-7)
        np.testing.assert_allclose(p.cpu().numpy(), pt.detach().numpy(), rtol=1e-5, atol=1e-6)
