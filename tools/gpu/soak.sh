# soak: 300 full-size iterations through the CLI (fused critic, hand-ordered forms, 64-row rollout) and 1500 iterations of the 512-env share
# (split critic on the second stream, 16-row rollout), at the reference's DEFAULT evaluation cadence (batched device evaluation on its own stream,
# round 4) and rollout logging: finite scalars, entropy moving, evaluation rounds all logged, no hang
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/soak
mkdir -p $O
cd $R
python - <<'P' 2>&1 | grep -v amdgpu | tee $O/soak.txt
import math, time, sys
sys.path.insert(0, ".")
from cleanmarl_amd.driver import run
for E, iters in ((4096, 300), (512, 1500)):
    T, A = 128, 8
    t0 = time.time()
    out = run("mappo_multienvs", ["--env_type=synthetic", f"--batch_size={E}", f"--synthetic_agents={A}", f"--synthetic_steps={T}",
                                  f"--total_timesteps={E * T * iters}", "--eval_steps=50", "--num_eval_ep=10", "--log_every=10",
                                  "--actor_hidden_dim=64", "--critic_hidden_dim=64"])
    h = out["history"]
    def series(tag): return [v for t, v, s in h if t == tag]
    al, cl, en, rw = series("train/actor_loss"), series("train/critic_loss"), series("train/entropy"), series("rollout/ep_reward")
    ev = series("eval/ep_reward")
    ok = all(math.isfinite(x) for x in al + cl + en + ev) and len(ev) == iters // 50
    print(f"E={E}: {iters} iterations in {time.time() - t0:.1f} s, finite={ok}, logged={len(al)}, entropy {en[0]:.4f} -> {en[-1]:.4f}, "
          f"critic_loss {cl[0]:.4f} -> {cl[-1]:.4f}, eval rounds {len(ev)} (eval/ep_reward {ev[0]:.2f} -> {ev[-1]:.2f}), ep_reward {rw[0] if rw else float('nan'):.2f} -> {rw[-1] if rw else float('nan'):.2f}")
P
