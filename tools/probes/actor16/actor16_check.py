"""Parity + timing of the opt-in wave-private actor kernel (CM_ACTOR_KERNEL=wave16) against k_mlp on identical inputs.
usage: python tools/probes/actor16_check.py out.pt [ref.pt]   (run once without and once with the environment variable; the second
run names the first run's file and prints the differences).  A16_SMALL=1 skips the config-3 shape and the timing loop."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from cleanmarl_amd import _native as N
from cleanmarl_amd.learner import DeviceBatch, HParams, NetSpec, PPOLearner, init_params_like_torch
if os.environ.get("A16_LIB"):
    N.LIB_PATH = os.environ["A16_LIB"]  # probe builds of the library (ablation variants)
dev = torch.device("cuda:0")
out = {}
SHAPES = [(37, 3, 25, 24, 5, 64), (4096, 8, 128, 56, 5, 64), (50, 2, 9, 64, 16, 48), (3, 1, 5, 8, 2, 32), (129, 4, 16, 56, 5, 64)]
if os.environ.get("A16_SMALL"):
    SHAPES = [s for s in SHAPES if s[0] != 4096]
for (E, A, T, Do, K, H) in SHAPES:
    torch.manual_seed(7)
    b = DeviceBatch(E, A, T, Do, 12, K, dev)
    b.obs.normal_(); b.state.normal_(); b.reward.normal_()
    b.avail.copy_((torch.rand(E, A, T, K, device=dev) < 0.7).to(torch.uint8)); b.avail[..., 0] = 1
    b.action.copy_(torch.zeros(E, A, T, dtype=torch.int32, device=dev))
    b.logp.copy_(-torch.rand(E, A, T, device=dev)); b.adv.normal_()
    b.ep_len.copy_(torch.randint(max(1, T // 2), T + 1, (E,), device=dev).int())
    aspec, cspec = NetSpec(Do, H, 1, K), NetSpec(12, 64, 1, 1)
    L = PPOLearner("mappo", aspec, cspec, A, HParams(), dev, init_params_like_torch(aspec), init_params_like_torch(cspec))
    s = N.stream_ptr()
    L.actor_pass(b, s)
    torch.cuda.synchronize()
    out[(E, A, T, Do, K, H)] = L.g_actor.clone().cpu()
    if E == 4096:
        for _ in range(100):
            L.actor_pass(b, s)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            L.actor_pass(b, s)
        e1.record(); torch.cuda.synchronize()
        print("actor pass at cfg 3:", round(e0.elapsed_time(e1) / 20, 4), "ms  kernel =", os.environ.get("CM_ACTOR_KERNEL", "k_mlp"), os.environ.get("A16_LIB", ""))
torch.save(out, sys.argv[1])
if len(sys.argv) > 2:
    ref = torch.load(sys.argv[2])
    for k, v in out.items():
        d = (v - ref[k]).abs() / (1 + ref[k].abs())
        print(k, "max rel err vs", sys.argv[2], float(d.max()), "stats", v[-8:].tolist()[:6], ref[k][-8:].tolist()[:6])
