import sys, time, torch
sys.path.insert(0, "/root/repo")
from cleanmarl_amd.learner import DeviceBatch, HParams, NetSpec, PPOLearner, init_params_like_torch
from cleanmarl_amd import _native as N
dev = torch.device("cuda:0")
E, A, T, K = 1024, 5, 128, 5
Do, Ds = 6 * A + A, 6 * A * A
for pad in (False, True):
    b = DeviceBatch(E, A, T, Do, Ds, K, dev, pad_obs=False, pad_state=pad)
    b.state.normal_(); b.obs.normal_(); b.avail.fill_(1); b.ep_len.fill_(T); b.reward.normal_()
    aspec, cspec = NetSpec(Do, 64, 1, K), NetSpec(Ds, 64, 1, 1)
    torch.manual_seed(0)
    L = PPOLearner("mappo", aspec, cspec, A, HParams(), dev, init_params_like_torch(aspec), init_params_like_torch(cspec))
    L._ensure_ws(b); L.compute_targets(b)
    s = N.stream_ptr()
    for _ in range(5): L.critic_pass(b, s)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): L.critic_pass(b, s)
    e1.record(); torch.cuda.synchronize()
    print("pad_state", pad, "state_ld", b.state_ld, "critic pass %.1f us" % (e0.elapsed_time(e1) * 1000 / 50))
