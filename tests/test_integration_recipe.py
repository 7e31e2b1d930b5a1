"""INTEGRATION.md sections 0-2 executed literally: raw ctypes + torch only (no cleanmarl_amd Python), binding the C-ABI at the
program regions of cleanmarl/mappo_multienvs.py it replaces, checked against a golden captured from the unmodified reference.
Proves the boundary is usable from the reference's own language without this repo's host code."""
import ctypes as C
import os

import numpy as np
import pytest
import torch

from parity import DISP_TOL, GRAD_TOL, TOL, StepChecker, check_grads, check_step, golden_before, golden_init, grad_err  # noqa: F401
from parity import err as _err

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", ["mappo_dense", "mappo_deep"])
def test_ctypes_recipe_reproduces_the_reference(golden_dir, name):
    z = np.load(os.path.join(golden_dir, name + ".npz"))
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    # ---- section 0
    lib = C.CDLL(os.path.join(root, "cleanmarl_amd", "libcleanmarl_hip.so"))
    lib.cm_last_error.restype = C.c_char_p
    for f in (lib.cm_mlp_train_workspace_bytes, lib.cm_critic_workspace_bytes):
        f.restype = C.c_size_t

    def chk(rc):
        if rc:
            raise RuntimeError(lib.cm_last_error().decode())
    P = lambda t: C.c_void_p(t.data_ptr())
    S = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
    d = C.c_double
    t = lambda k: torch.from_numpy(z[k])
    b_obs, b_actions, b_log_probs, b_reward = t("b_obs"), t("b_actions"), t("b_log_probs"), t("b_reward")
    b_states, b_avail_actions, b_mask = t("b_states"), t("b_avail_actions"), t("b_mask")
    E, T, A, Do = b_obs.shape
    Ds, K = b_states.shape[-1], b_avail_actions.shape[-1]
    obs = b_obs.permute(0, 2, 1, 3).contiguous().cuda()
    avail = b_avail_actions.permute(0, 2, 1, 3).to(torch.uint8).contiguous().cuda()
    action = b_actions.permute(0, 2, 1).to(torch.int32).contiguous().cuda()
    logp = b_log_probs.permute(0, 2, 1).contiguous().cuda()
    state, reward = b_states.cuda(), b_reward.cuda()
    ep_len = b_mask.sum(1).to(torch.int32).cuda()
    ap = [t(f"actor_init_{i}") for i in range(int(z["actor_nparam"]))]
    cp = [t(f"critic_init_{i}") for i in range(int(z["critic_nparam"]))]
    actor_flat = torch.cat([p.reshape(-1) for p in ap]).cuda()
    critic_flat = torch.cat([p.reshape(-1) for p in cp]).cuda()
    Ha, La, Hc, Lc = ap[0].shape[0], len(ap) // 2 - 2, cp[0].shape[0], len(cp) // 2 - 2
    hp = {k[3:]: (float(z[k]) if z[k].dtype.kind == "f" else str(z[k])) for k in z.files if k.startswith("hp_")}
    # ---- section 1
    values = torch.empty(E, 1, T, device="cuda")
    chk(lib.cm_mlp_forward(P(state), C.c_int64(E * T), Ds, Hc, Lc, 1, P(critic_flat), None, P(values), S()))
    ret, adv = torch.empty(E, A, T, device="cuda"), torch.empty(E, A, T, device="cuda")
    chk(lib.cm_td_lambda_scan(P(reward), P(values), P(ep_len), E, A, 1, T, d(hp["gamma"]), d(hp["td_lambda"]), P(ret), P(adv), S()))
    assert _err(ret.permute(0, 2, 1).cpu().numpy(), z["return_lambda"]) <= TOL
    assert _err(adv.permute(0, 2, 1).cpu().numpy(), z["advantages"]) <= TOL
    # ---- section 2
    Pa, Pc = actor_flat.numel(), critic_flat.numel()
    gbuf = torch.zeros(Pa + 8 + Pc + 8, device="cuda")
    ga, gc = gbuf[:Pa + 8], gbuf[Pa + 8:]
    ws = torch.empty(max(lib.cm_mlp_train_workspace_bytes(Do, Ha, La, K), lib.cm_critic_workspace_bytes(E, A, T, 0, Ds, Hc, Lc)),
                     dtype=torch.uint8, device="cuda")
    m_a, v_a, m_c, v_c = (torch.zeros(n, device="cuda") for n in (Pa, Pa, Pc, Pc))
    norms = torch.zeros(2, device="cuda")
    adamw = hp["optimizer"] == "AdamW"
    chk_a = StepChecker(golden_init(z, "actor"), str(hp["optimizer"]), float(hp["learning_rate_actor"]), "ctypes recipe actor")
    chk_c = StepChecker(golden_init(z, "critic"), str(hp["optimizer"]), float(hp["learning_rate_critic"]), "ctypes recipe critic")
    for epoch in range(int(hp["epochs"])):
        chk(lib.cm_ppo_actor_fwd_bwd(P(obs), P(avail), P(action), P(logp), P(adv), P(ep_len), E, A, T, Do, Ha, La, K, P(actor_flat),
                                     d(hp["ppo_clip"]), d(hp["entropy_coef"]), P(ga), P(ws), C.c_size_t(ws.numel()), S()))
        chk(lib.cm_critic_fwd_bwd(P(state), P(ret), P(ep_len), E, A, T, 0, Ds, Hc, Lc, P(critic_flat), P(gc), P(ws),
                                  C.c_size_t(ws.numel()), S()))
        step = epoch + 1
        for prm, g, m, v, lr, nrm in ((actor_flat, ga, m_a, v_a, hp["learning_rate_actor"], norms[0:]),
                                      (critic_flat, gc, m_c, v_c, hp["learning_rate_critic"], norms[1:])):
            chk(lib.cm_grad_norm_clip_adam(P(prm), P(g), P(m), P(v), C.c_int64(prm.numel()), step, d(lr), d(0.9), d(0.999), d(1e-8),
                                           d(0.01 if adamw else 0.0), 1 if adamw else 0, d(hp["clip_gradients"]), d(1.0), P(nrm), S()))
        st = ga[Pa:].cpu(); N = st[5]
        assert _err(float((-st[0] - hp["entropy_coef"] * st[1]) / N), z["actor_losses"][epoch]) <= TOL
        assert _err(float(st[1] / N), z["entropies_bonuses"][epoch]) <= TOL
        assert _err(float(st[2] / N), z["kl_divergences"][epoch]) <= TOL
        assert _err(float(st[3] / N), z["clipped_ratios"][epoch]) <= TOL
        assert _err(float(gc[Pc + 4] / gc[Pc + 5]), z["critic_losses"][epoch]) <= TOL
        assert grad_err(float(norms[0]), z["actor_gradients"][epoch]) <= GRAD_TOL and grad_err(float(norms[1]), z["critic_gradients"][epoch]) <= GRAD_TOL
        chk_a.step(ga[:Pa], actor_flat, z["actor_grads"][epoch], z["actor_after"][epoch])  # (the step leaves the gradient it consumed in the buffer)
        chk_c.step(gc[:Pc], critic_flat, z["critic_grads"][epoch], z["critic_after"][epoch])


class OptStep(C.Structure):  # cm_opt_step_t exactly as INTEGRATION.md section 6c declares it
    _fields_ = [("params", C.c_void_p), ("exp_avg", C.c_void_p), ("exp_avg_sq", C.c_void_p), ("out_norm", C.c_void_p), ("scratch", C.c_void_p),
                ("lr", C.c_double), ("beta1", C.c_double), ("beta2", C.c_double), ("eps", C.c_double), ("weight_decay", C.c_double),
                ("max_norm", C.c_double), ("grad_scale", C.c_double), ("step", C.c_int32), ("opt_kind", C.c_int32), ("stats_out", C.c_void_p)]


@pytest.mark.parametrize("name", ["mappo_dense", "mappo_ragged_norm"])
def test_ctypes_train_step_recipe_reproduces_the_reference(golden_dir, name):
    """INTEGRATION.md section 6c with nothing but ctypes: cm_set_option / cm_get_option, cm_opt_step_t, the *_train_step_ld entry points
    (the optimiser step rides on the pass's reduction launch) -- per-epoch parameters of the unmodified reference."""
    z = np.load(os.path.join(golden_dir, name + ".npz"))
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib = C.CDLL(os.path.join(root, "cleanmarl_amd", "libcleanmarl_hip.so"))
    lib.cm_last_error.restype = C.c_char_p
    lib.cm_get_option.restype = C.c_char_p
    for f in (lib.cm_mlp_train_workspace_bytes, lib.cm_critic_workspace_bytes, lib.cm_opt_step_scratch_bytes):
        f.restype = C.c_size_t

    def chk(rc):
        if rc:
            raise RuntimeError(lib.cm_last_error().decode())
    assert lib.cm_set_option(b"critic_schedule", b"nonsense") != 0 and lib.cm_set_option(b"no_such_option", b"auto") != 0
    chk(lib.cm_set_option(b"critic_schedule", b"split"))
    assert lib.cm_get_option(b"critic_schedule") == b"split" and lib.cm_get_option(b"mfma") == b"fp32"
    P = lambda t: C.c_void_p(t.data_ptr())
    S = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
    d = C.c_double
    t = lambda k: torch.from_numpy(z[k])
    b_obs, b_states, b_avail, b_mask = t("b_obs"), t("b_states"), t("b_avail_actions"), t("b_mask")
    E, T, A, Do = b_obs.shape
    Ds, K = b_states.shape[-1], b_avail.shape[-1]
    obs = b_obs.permute(0, 2, 1, 3).contiguous().cuda()
    avail = b_avail.permute(0, 2, 1, 3).to(torch.uint8).contiguous().cuda()
    action = t("b_actions").permute(0, 2, 1).to(torch.int32).contiguous().cuda()
    logp = t("b_log_probs").permute(0, 2, 1).contiguous().cuda()
    state = b_states.cuda()
    ep_len = b_mask.sum(1).to(torch.int32).cuda()
    # targets as the reference computed them (sections 1 / 1b are covered by the test above): the [E, A, T] permutation of the goldens
    ret = t("return_lambda").permute(0, 2, 1).contiguous().cuda()
    adv = t("advantages").permute(0, 2, 1).contiguous().cuda()
    ap = [t(f"actor_init_{i}") for i in range(int(z["actor_nparam"]))]
    cp = [t(f"critic_init_{i}") for i in range(int(z["critic_nparam"]))]
    actor_flat = torch.cat([p.reshape(-1) for p in ap]).cuda()
    critic_flat = torch.cat([p.reshape(-1) for p in cp]).cuda()
    Ha, La, Hc, Lc = ap[0].shape[0], len(ap) // 2 - 2, cp[0].shape[0], len(cp) // 2 - 2
    hp = {k[3:]: (float(z[k]) if z[k].dtype.kind == "f" else str(z[k])) for k in z.files if k.startswith("hp_")}
    Pa, Pc = actor_flat.numel(), critic_flat.numel()
    ga, gc = torch.zeros(Pa + 8, device="cuda"), torch.zeros(Pc + 8, device="cuda")
    ws = torch.empty(max(lib.cm_mlp_train_workspace_bytes(Do, Ha, La, K), lib.cm_critic_workspace_bytes(E, A, T, 0, Ds, Hc, Lc)),
                     dtype=torch.uint8, device="cuda")
    m_a, v_a, m_c, v_c = (torch.zeros(n, device="cuda") for n in (Pa, Pa, Pc, Pc))
    norms = torch.zeros(2, device="cuda")
    rec = torch.zeros(int(hp["epochs"]), 8, device="cuda")  # cm_opt_step_t::stats_out (ABI 101): the actor's statistic sums of every epoch
    assert lib.cm_version() >= 101
    scr_a, scr_c = (torch.zeros(lib.cm_opt_step_scratch_bytes(), dtype=torch.uint8, device="cuda") for _ in range(2))
    chk_a = StepChecker(golden_init(z, "actor"), str(hp["optimizer"]), float(hp["learning_rate_actor"]), "ctypes recipe actor")
    chk_c = StepChecker(golden_init(z, "critic"), str(hp["optimizer"]), float(hp["learning_rate_critic"]), "ctypes recipe critic")
    try:
        for epoch in range(int(hp["epochs"])):
            oa = OptStep(actor_flat.data_ptr(), m_a.data_ptr(), v_a.data_ptr(), norms.data_ptr(), scr_a.data_ptr(), hp["learning_rate_actor"], 0.9,
                         0.999, 1e-8, 0.0, hp["clip_gradients"], 1.0, epoch + 1, 0, rec[epoch].data_ptr())
            oc = OptStep(critic_flat.data_ptr(), m_c.data_ptr(), v_c.data_ptr(), norms[1:].data_ptr(), scr_c.data_ptr(), hp["learning_rate_critic"],
                         0.9, 0.999, 1e-8, 0.0, hp["clip_gradients"], 1.0, epoch + 1, 0, None)
            chk(lib.cm_ppo_actor_train_step_ld(P(obs), C.c_int64(Do), P(avail), P(action), P(logp), P(adv), P(ep_len), E, A, T, Do, Ha, La, K,
                                               d(hp["ppo_clip"]), d(hp["entropy_coef"]), P(ga), P(ws), C.c_size_t(ws.numel()), C.byref(oa), S()))
            chk(lib.cm_critic_train_step_ld(P(state), C.c_int64(Ds), P(ret), P(ep_len), E, A, T, 0, Ds, Hc, Lc, P(gc), P(ws),
                                            C.c_size_t(ws.numel()), C.byref(oc), S()))
            st = ga[Pa:].cpu(); N = st[5]
            assert torch.equal(rec[epoch].cpu(), st)  # the same sums, also where the caller asked for them
            assert _err(float((-st[0] - hp["entropy_coef"] * st[1]) / N), z["actor_losses"][epoch]) <= TOL
            assert _err(float(gc[Pc + 4] / gc[Pc + 5]), z["critic_losses"][epoch]) <= TOL
            assert grad_err(float(norms[0]), z["actor_gradients"][epoch]) <= GRAD_TOL and grad_err(float(norms[1]), z["critic_gradients"][epoch]) <= GRAD_TOL
            chk_a.step(ga[:Pa], actor_flat, z["actor_grads"][epoch], z["actor_after"][epoch])  # (the step leaves the gradient it consumed in the buffer)
            chk_c.step(gc[:Pc], critic_flat, z["critic_grads"][epoch], z["critic_after"][epoch])
    finally:
        chk(lib.cm_set_option(b"critic_schedule", b"auto"))
