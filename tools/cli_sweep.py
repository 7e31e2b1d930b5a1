"""Run every drop-in script with unusual flag combinations (normalisations, AdamW, clipping, 0 / 2 hidden layers, no agent ids,
pipe / shm vector envs, single env, batch_size 1, COMA n-step / target-update cadence) and check that it finishes with finite
scalars.  usage (on a GPU box): python tools/cli_sweep.py"""
import math, os, sys, tempfile, traceback
sys.path.insert(0, os.getcwd())
from cleanmarl_amd.driver import run
from cleanmarl_amd.coma_driver import run as run_coma
os.chdir(tempfile.mkdtemp())
base = ["--batch_size=5", "--synthetic_agents=4", "--synthetic_steps=13", "--total_timesteps=130", "--eval_steps=1", "--num_eval_ep=1", "--log_every=1"]
cases = [
 ("mappo_multienvs", ["--env_type=synthetic", "--no-agent_ids", "--normalize_reward", "--normalize_advantage", "--normalize_return", "--optimizer=AdamW", "--clip_gradients=0.5", "--actor_num_layers=2", "--critic_num_layers=0", "--actor_hidden_dim=48"]),
 ("ippo_multienvs", ["--env_type=synthetic_cpu", "--vector_env=pipe", "--agent_ids=False", "--normalize_advantage=True", "--epochs=1", "--critic_num_layers=2"]),
 ("mappo_lstm_multienvs", ["--env_type=synthetic", "--tbptt=4", "--normalize_return", "--clip_gradients=1.0", "--actor_hidden_dim=64"]),
 ("ippo_lstm_multienvs", ["--env_type=synthetic_shape", "--synthetic_obs=20", "--synthetic_state=12", "--synthetic_actions=6", "--tbptt=20", "--no-agent_ids"]),
 ("mappo", ["--env_type=synthetic_shape_cpu", "--synthetic_obs=9", "--synthetic_state=7", "--synthetic_actions=3", "--normalize_reward"]),
 ("ippo_lstm", ["--env_type=synthetic_cpu", "--greedy_eval", "--tbptt=3"]),
 ("mappo_multienvs", ["--env_type=synthetic", "--batch_size=1", "--synthetic_agents=1", "--synthetic_steps=2", "--total_timesteps=4"]),
 # the layered schedule (hidden 65..256 / deeper than 2 hidden layers)
 ("mappo_multienvs", ["--env_type=synthetic", "--critic_hidden_dim=128", "--normalize_advantage"]),
 ("ippo_multienvs", ["--env_type=synthetic_shape", "--actor_hidden_dim=128", "--critic_hidden_dim=96", "--critic_num_layers=3", "--synthetic_obs=20", "--synthetic_state=12", "--synthetic_actions=6"]),
 ("mappo", ["--env_type=synthetic_cpu", "--actor_num_layers=4", "--greedy_eval", "--clip_gradients=0.5"]),
 ("mappo_lstm_multienvs", ["--env_type=synthetic", "--critic_hidden_dim=256", "--tbptt=5"]),
 ("ippo_multienvs", ["--env_type=synthetic", "--batch_size=1", "--synthetic_agents=2", "--synthetic_steps=3", "--total_timesteps=6", "--normalize_advantage"]),
]
coma_cases = [
 ("coma_multienvs", ["--env_type=synthetic", "--critic_hidden_dim=64", "--use_tdlamda=False", "--nsteps=5", "--normalize_return", "--clip_gradients=0.5", "--no-agent_ids", "--target_network_update_freq=2", "--exploration_fraction=2"]),
 ("coma", ["--env_type=synthetic_shape_cpu", "--critic_hidden_dim=32", "--critic_num_layers=2", "--actor_num_layers=0", "--synthetic_obs=9", "--synthetic_state=7", "--synthetic_actions=3", "--normalize_advantage=False", "--optimizer=AdamW"]),
 ("coma_multienvs", ["--env_type=synthetic_cpu", "--critic_hidden_dim=64", "--vector_env=pipe", "--batch_size=1", "--synthetic_agents=1"]),
 ("coma_multienvs", ["--env_type=synthetic", "--clip_gradients=0.5"]),  # reference default critic width 128
 ("coma", ["--env_type=synthetic_shape", "--actor_hidden_dim=128", "--critic_hidden_dim=200", "--critic_num_layers=3", "--synthetic_obs=9", "--synthetic_state=7", "--synthetic_actions=3"]),
]
bad = 0
for script, extra in cases + coma_cases:
    args = [a for a in base if not any(a.split("=")[0] == e.split("=")[0] for e in extra)] + extra
    try:
        out = (run_coma if script.startswith("coma") else run)(script, args)
        fin = all(math.isfinite(v) for _, v, _ in out["history"])
        print(("OK  " if fin else "NaN "), script, " ".join(extra)[:110], "steps", out["step"], "updates", out["training_step"])
        bad += not fin
    except Exception as e:
        bad += 1
        print("FAIL", script, " ".join(extra)[:110]); traceback.print_exc(limit=3)
print("bad =", bad)
