# long soak through the CLI at the final kernels: thousands of iterations per script at the reference's default evaluation cadence,
# every logged scalar finite, every evaluation round logged, no hang (each leg under its own timeout)
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/soak3
mkdir -p $O
cd $R
timeout 1200 python - <<'P' 2>&1 | grep -v amdgpu | tee $O/soak_long.txt
import math, time, sys
sys.path.insert(0, ".")
from cleanmarl_amd.driver import run
from cleanmarl_amd.coma_driver import run as run_coma
def series(h, tag): return [v for t, v, s in h if t == tag]
legs = [("mappo_multienvs", run, 4096, 8, 128, 2000, []), ("mappo_multienvs", run, 512, 8, 128, 8000, []), ("ippo_multienvs", run, 1024, 3, 128, 6000, []),
        ("mappo_lstm_multienvs", run, 1024, 5, 128, 2500, []), ("ippo_lstm_multienvs", run, 256, 3, 64, 3000, []),
        ("coma_multienvs", run_coma, 4096, 8, 128, 800, ["--critic_hidden_dim=128"]), ("coma_multienvs", run_coma, 1024, 3, 128, 3000, [])]
for script, fn, E, A, T, iters, extra in legs:
    t0 = time.time()
    out = fn(script, ["--env_type=synthetic", f"--batch_size={E}", f"--synthetic_agents={A}", f"--synthetic_steps={T}",
                      f"--total_timesteps={E * T * iters}", "--eval_steps=50", "--num_eval_ep=10", "--log_every=50"] + extra)
    h = out["history"]
    ok = all(math.isfinite(float(v)) for _, v, _ in h)
    ev = series(h, "eval/ep_reward"); rw = series(h, "rollout/ep_reward")
    dt = time.time() - t0
    print(f"{script} {E} x {A} x {T} {' '.join(extra)}: {iters} iterations in {dt:.1f} s ({1e3 * dt / iters:.2f} ms each incl. start-up), finite={ok}, "
          f"eval rounds {len(ev)} of {iters // 50}, ep_reward {rw[0] if rw else float('nan'):.1f} -> {rw[-1] if rw else float('nan'):.1f}", flush=True)
P
