import os, sys, tempfile
sys.path.insert(0, os.getcwd())
from cleanmarl_amd.driver import run
os.chdir(tempfile.mkdtemp())
for script in ("mappo_lstm_multienvs", "ippo_lstm_multienvs"):
    out = run(script, ["--env_type=synthetic", "--batch_size=256", "--synthetic_agents=3", "--synthetic_steps=25", "--total_timesteps=960000",
                       "--eval_steps=100000", "--log_every=1", "--actor_hidden_dim=64", "--tbptt=10"])
    r = [v for t, v, _ in out["history"] if t == "rollout/ep_reward"]
    k = max(1, len(r) // 10)
    print(script, "iterations", len(r), "first10%", sum(r[:k]) / k, "last10%", sum(r[-k:]) / k)
