# soak of the late-round changes through the CLI: COMA at the reference's default 128-wide critic (transposed forward tile, pipelined gather,
# S + z0 GEMM epilogue, packed W0) and at 64 units, MAPPO-GRU with the padded state buffer and two deferred critic epochs: finite scalars, no hang
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/soak2
mkdir -p $O
cd $R
python - <<'P' 2>&1 | grep -v amdgpu | tee $O/soak_coma_gru.txt
import math, time, sys
sys.path.insert(0, ".")
from cleanmarl_amd.driver import run
from cleanmarl_amd.coma_driver import run as run_coma
def series(h, tag): return [v for t, v, s in h if t == tag]
for H, iters in ((128, 120), (64, 200)):
    E, T, A = 4096, 128, 8
    t0 = time.time()
    out = run_coma("coma_multienvs", ["--env_type=synthetic", f"--batch_size={E}", f"--synthetic_agents={A}", f"--synthetic_steps={T}",
                                       f"--total_timesteps={E * T * iters}", "--eval_steps=50", "--num_eval_ep=10", "--log_every=10",
                                       f"--critic_hidden_dim={H}"])
    h = out["history"]
    tags = sorted({t for t, _, _ in h})
    vals = [v for _, v, _ in h]
    ok = all(math.isfinite(float(x)) for x in vals)
    cl = series(h, "train/critic_loss"); al = series(h, "train/actor_loss"); ev = series(h, "eval/ep_reward")
    print(f"COMA critic {H}: {iters} iterations in {time.time() - t0:.1f} s, finite={ok}, tags={len(tags)}, critic_loss {cl[0]:.4f} -> {cl[-1]:.4f}, actor_loss {al[0]:.4f} -> {al[-1]:.4f}, eval rounds {len(ev)}")
for E, iters in ((1024, 300),):
    T, A = 128, 5
    t0 = time.time()
    out = run("mappo_lstm_multienvs", ["--env_type=synthetic", f"--batch_size={E}", f"--synthetic_agents={A}", f"--synthetic_steps={T}",
                                       f"--total_timesteps={E * T * iters}", "--eval_steps=50", "--num_eval_ep=10", "--log_every=10"])
    h = out["history"]
    vals = [v for _, v, _ in h]
    ok = all(math.isfinite(float(x)) for x in vals)
    cl = series(h, "train/critic_loss"); en = series(h, "train/entropy"); ev = series(h, "eval/ep_reward"); rw = series(h, "rollout/ep_reward")
    print(f"MAPPO-GRU E={E}: {iters} iterations in {time.time() - t0:.1f} s, finite={ok}, critic_loss {cl[0]:.4f} -> {cl[-1]:.4f}, entropy {en[0]:.4f} -> {en[-1]:.4f}, eval rounds {len(ev)} ({ev[0]:.2f} -> {ev[-1]:.2f}), ep_reward {rw[0]:.2f} -> {rw[-1]:.2f}")
P
