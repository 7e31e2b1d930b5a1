"""CPU sanity run of COMA with the ORACLE math (reference restatement) on the CPU twin of the synthetic env: does the episode
return improve or deteriorate?  (Not a test; a one-off check of whether the deterioration seen on the GPU is inherent.)"""
import sys, numpy as np, torch
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import coma as C, restatement as R
from cleanmarl_amd.env.synthetic import SyntheticSpreadEnv
from cleanmarl_amd.learner import NetSpec, init_params_like_torch
from cleanmarl_amd.coma_learner import coma_critic_input_dim
torch.manual_seed(1); np.random.seed(1)
E, A, T, K = 64, 3, 25, 5
envs = [SyntheticSpreadEnv(n_agents=A, agent_ids=True, max_cycles=T, seed=1, env_index=i) for i in range(E)]
Do, Ds = envs[0].get_obs_size(), envs[0].get_state_size()
ap = init_params_like_torch(NetSpec(Do, 64, 1, K)); cp = init_params_like_torch(NetSpec(coma_critic_input_dim(Do, Ds, A, K), 64, 1, K))
tp = [p.clone() for p in cp]
hp = dict(gamma=0.99, td_lambda=0.8, normalize_reward=0.0, normalize_advantage=1.0, normalize_return=0.0, target_network_update_freq=1.0,
          polyak=0.005, entropy_coef=1e-3, use_tdlamda=1.0, nsteps=1.0, clip_gradients=-1.0, optimizer="Adam", learning_rate_actor=5e-4,
          learning_rate_critic=5e-4)
oa, oc = R.AdamState(ap, 5e-4, "Adam"), R.AdamState(cp, 5e-4, "Adam")
ts = 0; hist = []
for it in range(int(sys.argv[1]) if len(sys.argv) > 1 else 150):
    eps = max((0.002 - 0.5) / 100 * ts + 0.5, 0.002)
    obs = np.stack([e.reset()[0] for e in envs]); B = dict(obs=[], states=[], avail=[], actions=[], reward=[])
    for t in range(T):
        st = np.stack([e.get_state() for e in envs]); av = np.stack([e.get_avail_actions() for e in envs]).astype(bool)
        with torch.no_grad():
            lg = R.actor_logits(ap, torch.from_numpy(obs).float(), torch.from_numpy(av))
            pr = (1 - eps) * torch.softmax(lg, -1) + torch.from_numpy(av).float() * (eps / torch.from_numpy(av).float().sum(-1, keepdim=True))
            act = torch.multinomial(pr.reshape(-1, K), 1).reshape(E, A)
        nobs, rew = [], []
        for i, e in enumerate(envs):
            o, r, d, tr, _ = e.step(act[i].numpy()); nobs.append(o); rew.append(r)
        B["obs"].append(obs); B["states"].append(st); B["avail"].append(av); B["actions"].append(act.numpy()); B["reward"].append(rew)
        obs = np.stack(nobs)
    batch = dict(obs=torch.tensor(np.stack(B["obs"], 1)).float(), states=torch.tensor(np.stack(B["states"], 1)).float(),
                 avail=torch.tensor(np.stack(B["avail"], 1)), actions=torch.tensor(np.stack(B["actions"], 1)).long(),
                 reward=torch.tensor(np.array(B["reward"]).T).float(), mask=torch.ones(E, T, dtype=torch.bool))
    hist.append(float(batch["reward"].sum(1).mean()))
    rec = C.update(ap, cp, tp, batch, hp, oa, oc, ts); ts = rec["training_step"]
k = max(1, len(hist) // 10)
print("oracle COMA on CPU env: first10%", sum(hist[:k]) / k, "last10%", sum(hist[-k:]) / k, "min", min(hist), "max", max(hist))
