#!/usr/bin/env python3
"""Per-phase cycle breakdown of the fused MLP training kernels (needs the -DCM_PHASE_PROF build).
    hipcc ... -DCM_PHASE_PROF -o cleanmarl_amd/libcleanmarl_hip_prof.so ; python tools/phase_prof.py [actor|critic]"""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cleanmarl_amd import _native as N  # noqa: E402

N.LIB_PATH = os.environ.get("CM_PROF_LIB", os.path.join(ROOT, "cleanmarl_amd", "libcleanmarl_hip_prof.so"))
lib = N.load()
lib.cm_prof_set_buffer.argtypes = [C.c_void_p]
which = sys.argv[1] if len(sys.argv) > 1 else "actor"
E, A, T, K = 4096, 8, 128, 5
if len(sys.argv) > 3:  # python tools/phase_prof.py rollout 1024 3
    E, A = int(sys.argv[2]), int(sys.argv[3])
Do, Ds = 7 * A, 6 * A * A
dev = torch.device("cuda:0")
from cleanmarl_amd.learner import NetSpec, flatten_params, init_params_like_torch  # noqa: E402
torch.manual_seed(0)
prof = torch.zeros(256, 16, dtype=torch.int64, device=dev)
lib.cm_prof_set_buffer(C.c_void_p(prof.data_ptr()))
ep_len = torch.full((E,), T, dtype=torch.int32, device=dev)
s = N.stream_ptr()
if which == "actor":
    spec = NetSpec(Do, 64, 1, K)
    p = flatten_params(init_params_like_torch(spec), dev)
    obs = torch.randn(E, A, T, Do, device=dev); avail = torch.ones(E, A, T, K, dtype=torch.uint8, device=dev)
    act = torch.randint(0, K, (E, A, T), dtype=torch.int32, device=dev); lp = -torch.rand(E, A, T, device=dev) - 1.0
    adv = torch.randn(E, A, T, device=dev)
    g = torch.zeros(spec.nparams + 8, device=dev)
    ws = torch.empty(lib.cm_mlp_train_workspace_bytes(Do, 64, 1, K), dtype=torch.uint8, device=dev)
    run = lambda: N.check(lib.cm_ppo_actor_fwd_bwd(N.ptr(obs), N.ptr(avail), N.ptr(act), N.ptr(lp), N.ptr(adv), N.ptr(ep_len),
                                                  E, A, T, Do, 64, 1, K, N.ptr(p), 0.2, 1e-3, N.ptr(g), N.ptr(ws), ws.numel(), s), "actor")
elif which == "gru":  # one TBPTT chunk of config 5 through the second-generation sweeps (cm_gru_v2.h)
    E, A, T, K = 1024, 5, 128, 5
    if len(sys.argv) > 3:
        E, A = int(sys.argv[2]), int(sys.argv[3])
    Do = 7 * A
    prof = torch.zeros(1024, 16, dtype=torch.int64, device=dev)
    lib.cm_prof_set_buffer(C.c_void_p(prof.data_ptr()))
    spec = NetSpec(Do, 64, 0, K, "gru")
    p = flatten_params(init_params_like_torch(spec), dev)
    obs = torch.randn(E, A, T, Do, device=dev); avail = torch.ones(E, A, T, K, dtype=torch.uint8, device=dev)
    act = torch.randint(0, K, (E, A, T), dtype=torch.int32, device=dev); lp = -torch.rand(E, A, T, device=dev) - 1.0
    adv = torch.randn(E, A, T, device=dev)
    ep_len = torch.full((E,), T, dtype=torch.int32, device=dev)
    g = torch.zeros(spec.nparams + 8, device=dev)
    ws = torch.empty(lib.cm_gru_workspace_bytes(E, A, Do, 64, K, 10), dtype=torch.uint8, device=dev)
    h0 = torch.zeros(E * A, 64, device=dev); h1 = torch.zeros(E * A, 64, device=dev)
    run = lambda: N.check(lib.cm_gru_actor_chunk_fwd_bwd(N.ptr(obs), N.ptr(avail), N.ptr(act), N.ptr(lp), N.ptr(adv), N.ptr(ep_len), E, A, T, 10, 20, Do,
                                                        64, K, N.ptr(p), N.ptr(h0), N.ptr(h1), 0.2, 1e-3, N.ptr(g), N.ptr(ws), ws.numel(), s), "gru chunk")
elif which == "act":  # the whole-episode act pass of config 4 (cm_policy_act_episode_ld: 115-wide obs padded to 116, 17 actions, avail masks)
    E, A, T, K = 2048, 10, 256, 17
    Do, ld = 115, 116
    spec = NetSpec(Do, 64, 1, K)
    p = flatten_params(init_params_like_torch(spec), dev)
    obs = torch.zeros(E, A, T, ld, device=dev); obs[..., :Do] = torch.randn(E, A, T, Do, device=dev)
    avail = (torch.rand(E, A, T, K, device=dev) < 0.7).to(torch.uint8); avail[..., 0] = 1
    act = torch.zeros(E, A, T, dtype=torch.int32, device=dev); lp = torch.zeros(E, A, T, device=dev)
    w0b = lib.cm_w0_image_bytes(Do, 64)
    w0 = torch.empty(max(16, w0b), dtype=torch.uint8, device=dev)
    run = lambda: N.check(lib.cm_policy_act_episode_ld(N.ptr(obs), ld, N.ptr(avail), E * A, T, Do, 64, 1, K, N.ptr(p), 7, 0, N.ptr(act), N.ptr(lp),
                                                      N.ptr(w0), w0b, s), "act")
elif which == "fused128":  # csrc/cm_mlp_fused128.h: the COMA critic's training pass at the reference's default width (obs block + z0 addend left out: cm_qcritic_fwd_bwd)
    spec = NetSpec(Do, 128, 1, K)
    p = flatten_params(init_params_like_torch(spec), dev)
    obs = torch.randn(E, A, T, Do, device=dev)
    act = torch.randint(0, K, (E, A, T), dtype=torch.int32, device=dev); tgt = torch.randn(E, A, T, device=dev)
    g = torch.zeros(spec.nparams + 8, device=dev)
    ws = torch.empty(lib.cm_mlp_split_workspace_bytes(E * A * T, Do, 128, 1, K), dtype=torch.uint8, device=dev)
    run = lambda: N.check(lib.cm_qcritic_fwd_bwd(N.ptr(obs), N.ptr(act), N.ptr(tgt), N.ptr(ep_len), E, A, T, Do, 128, 1, K, N.ptr(p), N.ptr(g),
                                                 N.ptr(ws), ws.numel(), s), "qcritic 128")
elif which == "grurollout":  # the fused GRU rollout of config 5 (k_gru32_rollout)
    from cleanmarl_amd.gru import GRUSyntheticRollout
    E, A, T, K = 1024, 5, 128, 5
    Do = 6 * A + A
    prof = torch.zeros(1024, 16, dtype=torch.int64, device=dev)
    lib.cm_prof_set_buffer(C.c_void_p(prof.data_ptr()))
    spec = NetSpec(Do, 64, 0, K, "gru")
    p = flatten_params(init_params_like_torch(spec), dev)
    roll = GRUSyntheticRollout(E, A, T, seed=1, device=dev, agent_ids=True)
    run = lambda: roll.collect(p, spec, fused=True)
elif which == "rollout":
    from cleanmarl_amd.rollout import SyntheticSpreadRollout
    prof = torch.zeros(1024, 16, dtype=torch.int64, device=dev)
    lib.cm_prof_set_buffer(C.c_void_p(prof.data_ptr()))
    spec = NetSpec(Do, 64, 1, K)
    p = flatten_params(init_params_like_torch(spec), dev)
    roll = SyntheticSpreadRollout(E, A, T, seed=1, device=dev)
    run = lambda: roll.collect(p, spec, fused=True)
else:
    spec = NetSpec(Ds, 64, 1, 1)
    p = flatten_params(init_params_like_torch(spec), dev)
    st = torch.randn(E, T, Ds, device=dev); ret = torch.randn(E, A, T, device=dev)
    g = torch.zeros(spec.nparams + 8, device=dev)
    ws = torch.empty(lib.cm_critic_workspace_bytes(E, A, T, 0, Ds, 64, 1), dtype=torch.uint8, device=dev)
    run = lambda: N.check(lib.cm_critic_fwd_bwd(N.ptr(st), N.ptr(ret), N.ptr(ep_len), E, A, T, 0, Ds, 64, 1, N.ptr(p),
                                                N.ptr(g), N.ptr(ws), ws.numel(), s), "critic")
for _ in range(int(os.environ.get("CM_PROF_WARMUP", "2"))):  # CM_PROF_WARMUP=300: steady-state clock (the GPU ramps for ~0.1 s after idle)
    run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); run(); e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1)
ph = prof.double().mean(0).cpu()
tot = float(ph.sum())
if which == "gru":
    CL = 10
    fwd4 = ["barrier_top", "gru2_step (fc1, 6 products, gates)", "-", "-", "-", "workspace stores",
            "head: barrier", "head: load h'", "head: logits", "head: ppo math", "head: dW2 + dh_head + store"]
    # eight-wave forward: thread 0 (recurrence) in slots 0..3, thread 256 (helpers) in slots 8..12; all per real step
    fwd8 = ["r: wait top", "r: gru2_step (fc1, 6 products, gates)", "r: obs tile", "r: two drain steps", "-", "-", "-", "-",
            "h: wait top", "h: I0 4 tile stores, relu h'", "h: wait x1", "h: I1 item loads, 2 tile stores", "h: wait h'", "h: I0 ppo math (odd steps)",
            "h: I1 logits (even) | dW2 + dh_head (odd)"]
    tile_opt = os.environ.get("CM_GRU_TILE", "auto")
    eight = tile_opt != "32"
    nt = (E * A + 31) // 32
    if tile_opt == "auto" and nt + (nt + 1) // 2 <= 256:  # k_gru2_fwdx: chain workgroups in rows [0, nt), head workgroups behind them
        fwd8 = ["r: wait top", "r: gru2_step (fc1, 6 products, gates)", "r: obs tile", "r: drain step", "-", "-", "-", "-",
                "h: wait top", "h: I0 h' (agent scope), vmcnt(0), publish, 4 tile stores", "h: wait x1", "h: I1 x1 store", "h: wait h'"]
        rows = prof[nt:nt + (nt + 1) // 2].double()
        ph = rows.mean(0).cpu()
        print(f"k_gru2_fwdx head workgroups: {rows.shape[0]}, cycles per 2-step pass")
        for i, n in enumerate(["items, flags + h' rows (agent scope) of both steps", "barrier, HB", "barrier, logits", "barrier, ppo math", "barrier, dW2 + dh_head (2 barriers)"]):
            print(f"  {n:34s} {float(ph[i]) / (CL / 2):10.1f}")
        print(f"  total cycles/WG {float(ph[:8].sum()):.0f}")
        prof[nt:512] = 0
    if tile_opt == "auto" and nt + (nt + 1) // 2 <= 256:  # k_gru2_bwd<true>: weight-gradient workgroups behind the chain workgroups
        ng = min(256 - nt, nt)
        rows = prof[512 + nt:512 + nt + ng].double()
        ph = rows.mean(0).cpu()
        units = nt * (CL - 1) / ng
        print(f"k_gru2_bwd<true> weight-gradient workgroups: {rows.shape[0]}, cycles per (tile, step) unit, {units:.2f} units each")
        for i, n in enumerate(["wait for the unit's {dh_t, tag} words", "gate derivatives, barrier, tiles, next loads", "barrier", "five products"]):
            print(f"  {n:34s} {float(ph[i]) / units:10.1f}")
        print(f"  total cycles/WG {float(ph[:8].sum()):.0f}")
        allr = prof[512:512 + nt + ng].double().cpu()   # slots 6, 7: s_memrealtime (100 MHz, device-wide) at the start / end of the workgroup
        t0 = allr[:, 6].min()
        for nm, rr_ in (("chain", allr[:nt]), ("weight-gradient", allr[nt:])):
            print(f"  {nm:16s} start {float((rr_[:, 6] - t0).min()) / 100:7.2f} .. {float((rr_[:, 6] - t0).max()) / 100:7.2f} us   end {float((rr_[:, 7] - t0).min()) / 100:7.2f} .. {float((rr_[:, 7] - t0).max()) / 100:7.2f} us")
        prof[512:, 6:8] = 0
        prof[512 + nt:1024] = 0
    for name, base, per, names in (("k_gru2_fwd8/x" if eight else "k_gru2_fwd", 0, CL, fwd8 if eight else fwd4),
                                   ("k_gru2_bwd", 512, CL, ["barrier_top", "gate derivatives (s, registers) | weight gradients (s + 1)", "tile writes", "data path (3 blocks, reg B)", "dx1 / dh write"])):
        rows = prof[base:base + 512]
        used = rows[rows.sum(1) > 0].double()
        ph = used.mean(0).cpu(); tot = float(ph.sum())
        if names is fwd8:
            tot = float(ph[:8].sum())
        print(f"{name}: {used.shape[0]} workgroups, cycles per step (head phases: per 2-step pass) -- fwd+bwd launch pair {ms:.3f} ms")
        for i, n in enumerate(names):
            div = per if not n.startswith("head") else per / 2
            print(f"  {n:34s} {float(ph[i]) / div:10.1f}  {100 * float(ph[i]) / tot:5.1f}%")
        print(f"  total cycles/WG {tot:.0f}")
    sys.exit(0)
if which == "grurollout":
    used = prof[(prof.sum(1) > 0)]
    ph = used.double().mean(0).cpu(); tot = float(ph.sum())
    print(f"gru rollout: {ms:.3f} ms; cycles per (tile, step) per phase ({used.shape[0]} WGs):")
    for i, n in enumerate(["barrier top", "reward partials + obs build", "barrier", "buffer writes + philox", "gru2_step", "head logits", "barrier", "sample + physics"]):
        print(f"  {n:30s} {float(ph[i]) / T:10.1f}  {100 * float(ph[i]) / tot:5.1f}%")
    print(f"  total cycles/WG {tot:.0f}")
    sys.exit(0)
if which == "rollout":
    used = prof[(prof.sum(1) > 0)]
    ph = used.double().mean(0).cpu(); tot = float(ph.sum())
    names = ["barrier_top", "obs+reward_partials", "buffer_writes", "fwd_mlp", "head_logits(mfma)", "sample+physics"]
    tile = os.environ.get("CM_ROLLOUT_TILE") or ("16s" if (E + 16 // A - 1) // (16 // A) <= 256 else "16" if (E + 16 // A - 1) // (16 // A) <= 768 else "64s")
    if tile == "16":
        names = ["barrier_top", "reward_partials+obs", "buffer_writes+philox", "layer0", "layer1", "head(mfma)", "sample+physics"]
    if tile == "64s":
        names = ["c: wait B0", "c: obs build", "c: wait B1", "c: layer 0", "c: wait B2", "c: layer 1", "c: wait B3", "c: head+sample+physics",
                 "w: (loop)", "w: wait B0 + B1", "w: obs/state stores", "w: wait B2", "w: philox", "w: wait B3", "s: B1..B3 + wait B0", "s: reward + flush"]
    if tile == "16s":  # compute wave 0: slots 0..7, store wave: slots 8..15 (work / wait at the barrier that follows it)
        names = ["c: wait B0", "c: obs build", "c: wait B1", "c: layer 0", "c: wait B2", "c: layer 1", "c: wait B3", "c: head+sample+physics",
                 "w: (loop)", "w: wait B0", "w: philox", "w: wait B1", "w: obs/state stores", "w: wait B2 (+B3)", "-", "-"]
    print(f"rollout: {ms:.3f} ms; ticks per (tile,step) per phase ({used.shape[0]} WGs):")
    if tile in ("16s", "64s"):
        tot = float(ph[:8].sum())
    for i, n in enumerate(names):
        print(f"  {n:24s} {float(ph[i]) / T:10.1f}  {100 * float(ph[i]) / tot:5.1f}%")
    print(f"  total ticks/WG {tot:.0f}  -> {tot / (ms * 1e-3) / 1e6:.1f} MHz tick rate")
    sys.exit(0)
if which == "critic" and os.environ.get("CM_CRITIC_SCHEDULE") != "split" and 128 < Ds <= 448:  # csrc/cm_critic_fused.h
    names = ["fwd_L0 (X -> LDS, NC chunks)", "h0 -> LDS", "fwd_L1 + value", "loss", "dZ1 -> LDS", "dW1", "dH0 + dZ0 -> LDS", "dW0", "barrier_end"]
    tiles_per_wg = E * T / 64 / 256
    print(f"critic (fused): {ms:.3f} ms, {tiles_per_wg:.0f} tiles/WG, s_memtime ticks per tile:")
    for i, n in enumerate(names):
        print(f"  {n:30s} {float(ph[i]) / tiles_per_wg:10.1f}  {100 * float(ph[i]) / tot:5.1f}%")
    print(f"  total ticks/WG {tot:.0f}  -> {tot / (ms * 1e-3) / 1e6:.1f} MHz tick rate")
    sys.exit(0)
if which == "fused128":
    names = ["layer 0 (z0 loads, 64 MFMAs, H0 -> LDS)", "wait B1", "layer 1 (128 MFMAs, H1 -> LDS)", "wait B2", "head (32 MFMAs) + X request", "wait B3",
             "loss heads (wave 0)", "wait B4", "X -> LDS, dWout, dH1 (64 MFMAs) -> LDS", "dW1 (128 MFMAs)", "wait B5", "dH0 (128 MFMAs), dz0 stores", "dW0 (64 MFMAs)"]
    tiles_per_wg = E * A * T / 64 / 256
    print(f"fused 128-wide tile, M_QCRITIC, {E} x {A} x {T} rows: {ms:.3f} ms, {tiles_per_wg:.0f} tiles/WG, s_memtime ticks of thread 0 per tile:")
    for i, n in enumerate(names):
        print(f"  {n:44s} {float(ph[i]) / tiles_per_wg:10.1f}  {100 * float(ph[i]) / tot:5.1f}%")
    print(f"  total ticks/WG {tot:.0f}  -> {tot / (ms * 1e-3) / 1e6:.1f} MHz tick rate")
    sys.exit(0)
names = ["stage_x", "fwd_L0", "fwd_hidden", "head_logits", "softmax_loss", "dWout", "dZ_L", "bwd_hidden(colred+tn)", "inplace", "bwd_L0(colred)"]
if which == "act":
    names = ["stage x / W0 chunks + MFMAs of chunk 0", "fwd_L0 (last chunk) + epilogue", "fwd_hidden", "head_logits + mask", "philox + sampler + stores"]
rows = (E * A * T) if which in ("actor", "act") else E * T
tiles_per_wg = rows / 64 / 512  # two workgroups per CU
print(f"{which}: {ms:.3f} ms, {tiles_per_wg:.0f} tiles/WG, s_memtime ticks (100 MHz const clock?) per tile:")
for i, n in enumerate(names):
    print(f"  {n:24s} {float(ph[i]) / tiles_per_wg:10.1f}  {100 * float(ph[i]) / tot:5.1f}%")
print(f"  total ticks/WG {tot:.0f}  -> {tot / (ms * 1e-3) / 1e6:.1f} MHz tick rate")
